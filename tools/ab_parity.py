#!/usr/bin/env python3
"""Result of one solve as a file, for bitwise comparison between library builds / schedule options:
    [J2P_LIBRARY=...] python tools/ab_parity.py OUT.npy [opt=value ...]     (opt: fold, nip, ntg, mixed)
    python tools/ab_parity.py --cmp A.npy B.npy ..."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "--cmp":
    ref = np.load(sys.argv[2])
    bad = 0
    for f in sys.argv[3:]:
        a = np.load(f)
        same = a.shape == ref.shape and np.array_equal(a.view(np.uint32), ref.view(np.uint32))
        print(f"{f}: {'bit-identical' if same else 'DIFFERS'} vs {sys.argv[2]}")
        bad += not same
    sys.exit(1 if bad else 0)

import jpeg2png_amd as j  # noqa: E402
from jpeg2png_amd import synth  # noqa: E402

out = sys.argv[1]
opts = dict(kv.split("=") for kv in sys.argv[2:])
res = []
for (w, h, samp, yonly, its) in ((1000, 744, "444", True, 40), (4096, 1024, "444", True, 12), (640, 480, "420", False, 25)):
    planes = synth.make_planes(w, h, samp, 10, seed=77, y_only=yonly)
    s = j.Solver(planes, 0.3, [0.001] * len(planes), its)
    for name, oid in (("fold", j.J2P_OPT_NORM_FOLD), ("nip", j.J2P_OPT_NORM_IN_PROJECT), ("ntg", j.J2P_OPT_NT_GRADIENT), ("mixed", j.J2P_OPT_MIXED_PROJECT)):
        if name in opts:
            s.debug_option(oid, int(opts[name]))
    s.run(its)
    s.sync()
    for c in range(len(planes)):
        res.append(s.download(c).ravel())
    s.close()
np.save(out, np.concatenate(res))
print("wrote", out)
