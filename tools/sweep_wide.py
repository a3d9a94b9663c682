"""second randomised sweep (tests/sweep_cases.py: cases_wide): wider sizes, continuous weights, every quality,
16-bit-range quantisation tables, sparse coefficient data — GPU vs the compiled reference, bitwise.
usage: python tools/sweep_wide.py [ncases] [seed]"""
import copy
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import jpeg2png_amd as j
from oracle import bindings as oracle
from sweep_cases import cases_wide, planes_wide

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad = 0
t_ref = 0.0
for cs in cases_wide(seed, n):
    planes = planes_wide(cs)
    for p in planes:
        p.fdata = j.decode_plane(p)
    t0 = time.perf_counter()
    want, want_log, _ = oracle.ref_compute(planes, cs.weight, cs.pweights, cs.iterations, log=cs.log)
    t_ref += time.perf_counter() - t0
    got = copy.deepcopy(planes)
    got_log = j.compute(got, cs.weight, cs.pweights, cs.iterations, log=cs.log)
    same = all(np.array_equal(g.fdata.view(np.uint32), w.view(np.uint32)) for g, w in zip(got, want))
    logok = True
    if cs.log and cs.iterations:
        logok = np.allclose(got_log[:, 1:], want_log[:, 1:], rtol=1e-9, atol=2e-6 * max(1.0, float(np.abs(want_log[:, 1:]).max())))
    ok = same and logok
    bad += not ok
    extra = ""
    if not same:
        extra = "  max|d| " + str(max(float(np.abs(g.fdata - w).max()) for g, w in zip(got, want))) + " ndiff " + \
                str([int((g.fdata.view(np.uint32) != w.view(np.uint32)).sum()) for g, w in zip(got, want)])
    print(("ok   " if ok else "DIFF ") + cs.describe() + f" qscale {cs.qscale} sparse {int(cs.sparsify)}" + extra
          + ("" if logok else "  LOG"), flush=True)
print(f"{n - bad}/{n} bit-identical; reference {t_ref:.1f} s")
sys.exit(1 if bad else 0)
