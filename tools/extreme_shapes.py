"""extreme aspect ratios against the compiled reference — the JPEG limit (65500 pixels a side) in one dimension —
and two joint images of ~9.5 Mpixel"""
import copy
import sys

import numpy as np

sys.path.insert(0, ".")
import jpeg2png_amd as j
from jpeg2png_amd import synth
from oracle import bindings as oracle

bad = 0
for W, H, sub, y_only, its in [(65500, 16, "444", True, 6), (16, 65500, "444", True, 6), (65500, 24, "420", False, 5),
                               (24, 65500, "420", False, 5), (40000, 40, "422", False, 4), (33, 40000, "440", False, 4),
                               (65500, 8, "411", False, 3), (8, 65500, "410", False, 3),
                               (3072, 3200, "420", False, 4), (3100, 3050, "444", False, 3)]:
    planes = synth.make_planes(W, H, sub, 25, seed=W + H, y_only=y_only)
    for p in planes:
        p.fdata = j.decode_plane(p)
    pws = [0.001] * len(planes)
    want, want_log, _ = oracle.ref_compute(planes, 0.3, pws, its, log=True)
    got = copy.deepcopy(planes)
    got_log = j.compute(got, 0.3, pws, its, log=True)
    same = all(np.array_equal(g.fdata.view(np.uint32), w.view(np.uint32)) for g, w in zip(got, want))
    logok = np.allclose(got_log[:, 1:], want_log[:, 1:], rtol=1e-9, atol=2e-6)
    bad += not (same and logok)
    print(("ok   " if same and logok else "DIFF ") + f"{W}x{H} {sub} {'Y' if y_only else 'YCC'} its {its}", flush=True)
sys.exit(1 if bad else 0)
