"""nchannel == 2 (allowed by compute.c:118, never used by the CLI) against the compiled reference"""
import copy
import os
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import jpeg2png_amd as j
from oracle import bindings as oracle
from sweep_cases import cases

bad = n = 0
for cs in cases(77, 60):
    if cs.y_only:
        continue
    planes = cs.planes()
    pick = [(0, 1), (1, 2), (0, 2)][cs.index % 3]
    planes = [planes[pick[0]], planes[pick[1]]]
    for p in planes:
        p.fdata = j.decode_plane(p)
    pws = [cs.pweights[pick[0]], cs.pweights[pick[1]]]
    want, want_log, _ = oracle.ref_compute(planes, cs.weight, pws, cs.iterations, log=True)
    got = copy.deepcopy(planes)
    got_log = j.compute(got, cs.weight, pws, cs.iterations, log=True)
    same = all(np.array_equal(g.fdata.view(np.uint32), w.view(np.uint32)) for g, w in zip(got, want))
    if cs.iterations:
        same = same and np.allclose(got_log[:, 1:], want_log[:, 1:], rtol=1e-9, atol=2e-6)
    n += 1
    bad += not same
    print(("ok   " if same else "DIFF ") + cs.describe() + f" channels {pick}", flush=True)
print(f"{n - bad}/{n} two-channel cases bit-identical")
sys.exit(1 if bad else 0)
