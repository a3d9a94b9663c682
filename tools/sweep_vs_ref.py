"""randomised sweep: GPU solver vs the compiled reference (oracle/_ref), bitwise (so also the sign of zero).
usage: python tools/sweep_vs_ref.py [ncases] [seed]     SWEEP_ONLY=i,j reruns single cases
(test infrastructure; needs oracle/_ref; the case stream is tests/sweep_cases.py)"""
import copy
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import jpeg2png_amd as j
from oracle import bindings as oracle
from sweep_cases import cases

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = [int(x) for x in os.environ["SWEEP_ONLY"].split(",")] if os.environ.get("SWEEP_ONLY") else None
bad = ran = 0
t_ref = t_gpu = 0.0
for cs in cases(seed, n):
    if only is not None and cs.index not in only:
        continue
    ran += 1
    planes = cs.planes()
    for p in planes:
        p.fdata = j.decode_plane(p)
    t0 = time.perf_counter()
    want, want_log, _ = oracle.ref_compute(planes, cs.weight, cs.pweights, cs.iterations, log=cs.log)
    t1 = time.perf_counter()
    got = copy.deepcopy(planes)
    got_log = j.compute(got, cs.weight, cs.pweights, cs.iterations, log=cs.log)
    t2 = time.perf_counter()
    t_ref += t1 - t0
    t_gpu += t2 - t1
    same = all(np.array_equal(g.fdata.view(np.uint32), w.view(np.uint32)) for g, w in zip(got, want))
    logok = True
    if cs.log and cs.iterations:
        logok = np.allclose(got_log[:, 1:], want_log[:, 1:], rtol=1e-9, atol=2e-6)
    if not (same and logok):
        bad += 1
    print(("ok   " if same and logok else "DIFF ") + cs.describe()
          + ("" if same else "  max|d| " + str(max(float(np.abs(g.fdata - w).max()) for g, w in zip(got, want)))), flush=True)
    if not same:
        for c, (g, w) in enumerate(zip(got, want)):
            dmask = g.fdata.view(np.uint32) != w.view(np.uint32)
            ys, xs = np.nonzero(dmask)
            if len(ys):
                print(f"      channel {c} ({planes[c].w}x{planes[c].h} samp {planes[c].w_samp}x{planes[c].h_samp}): {len(ys)} differ, "
                      f"bbox x {xs.min()}..{xs.max()} y {ys.min()}..{ys.max()}; first:",
                      [(int(x), int(y), hex(int(g.fdata.view(np.uint32)[y, x])), hex(int(w.view(np.uint32)[y, x])))
                       for y, x in list(zip(ys, xs))[:6]], flush=True)
print(f"{ran - bad}/{ran} bit-identical; reference {t_ref:.1f} s, gpu path {t_gpu:.1f} s")
sys.exit(1 if bad else 0)
