"""diagnostic: where does the GPU solver first differ from the compiled reference on a synthetic case?
usage: python tools/diff_vs_ref.py W H SUB ITS [y_only]   (needs oracle/_ref; test infrastructure only)"""
import copy
import sys

import numpy as np

sys.path.insert(0, ".")
import jpeg2png_amd as j
from jpeg2png_amd import synth
from oracle import bindings as oracle

W, H, sub, its = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
y_only = len(sys.argv) > 5
planes = synth.make_planes(W, H, sub, 10, seed=4321, y_only=y_only)
for p in planes:
    p.fdata = j.decode_plane(p)
pws = [0.001] * len(planes)


def run(n):
    want, _, _ = oracle.ref_compute(planes, 0.3, pws, n)
    got = copy.deepcopy(planes)
    j.compute(got, 0.3, pws, n)
    return [g.fdata for g in got], want


lo, hi = 0, its
got, want = run(its)
if all(np.array_equal(g.view(np.uint32), w.view(np.uint32)) for g, w in zip(got, want)):
    print("identical at", its)
    sys.exit(0)
while hi - lo > 1:
    mid = (lo + hi) // 2
    got, want = run(mid)
    same = all(np.array_equal(g.view(np.uint32), w.view(np.uint32)) for g, w in zip(got, want))
    print("its", mid, "same" if same else "DIFF", flush=True)
    if same:
        lo = mid
    else:
        hi = mid
got, want = run(hi)
print("first differing iteration count:", hi)
for c, (g, w) in enumerate(zip(got, want)):
    d = g.view(np.uint32) != w.view(np.uint32)
    ys, xs = np.nonzero(d)
    if len(ys) == 0:
        print(" channel", c, "identical")
        continue
    print(" channel", c, "ndiff", d.sum(), "of", d.size, "bbox x", xs.min(), xs.max(), "y", ys.min(), ys.max(),
          "max|d|", np.abs(g - w).max(), "first", (int(xs[0]), int(ys[0])), float(g[ys[0], xs[0]]), float(w[ys[0], xs[0]]))
    big = np.abs(g - w) > 1e-4
    ys, xs = np.nonzero(big)
    print("   big diffs:", big.sum(), "bbox x", xs.min(), xs.max(), "y", ys.min(), ys.max())
    for yy, xx in list(zip(ys, xs))[:12]:
        print("    ", xx, yy, float(g[yy, xx]), float(w[yy, xx]))
got2, _ = run(hi)
print("GPU deterministic across runs:", all(np.array_equal(a, b) for a, b in zip(got, got2)))
