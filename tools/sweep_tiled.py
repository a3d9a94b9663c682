"""randomised sweep of the C row tiling (j2p_tiled) on ONE GPU: the case stream of tests/sweep_cases.py (sizes from a few
pixels to ~1400 x 1100, six samplings, TV-only, per-channel pweights incl. 0, flat areas, logging) cut at random aligned
rows into 2..6 bands, through a random exchange of the engine — direct with each of its three ways of waiting, copy with
either norm schedule — and iterations issued in two run() calls: planes bitwise and CSV rows against the whole-canvas
solver.  usage: python tools/sweep_tiled.py [ncases] [seed]"""
import copy
import os
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
# (the switches this tool turns live in the experiments build of the library)
if "J2P_LIBRARY" not in os.environ:
    from jpeg2png_amd.buildlib import build_experiments
    os.environ["J2P_LIBRARY"] = build_experiments()
import jpeg2png_amd as j
from jpeg2png_amd import tiled
from sweep_cases import cases

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng([seed, 4242])
MODES = [("direct", "all", "root"), ("direct", "root", "root"), ("direct", "collector", "root"), ("direct", "counter", "root"), ("copy", "all", "root"),
         ("copy", "all", "all")]
bad = ran = 0
used = {}
for cs in cases(seed, n):
    planes = cs.planes()
    for p in planes:
        p.fdata = j.decode_plane(p)
    align = tiled.band_alignment(planes)
    H = max(p.h * p.h_samp for p in planes)
    units = (H + align - 1) // align
    if units < 2:
        continue
    nb = int(rng.integers(2, min(6, units) + 1))
    cuts = [0] + sorted((rng.choice(np.arange(1, units), nb - 1, replace=False) * align).tolist()) + [H]
    ex, wait, norm = MODES[int(rng.integers(len(MODES)))]
    its = max(2, cs.iterations)
    ref = copy.deepcopy(planes)
    want_rows = j.compute(ref, cs.weight, cs.pweights, its, log=cs.log)
    os.environ["J2P_TILED_EXCHANGE"], os.environ["J2P_TILED_WAIT"], os.environ["J2P_TILED_NORM"] = ex, wait, norm
    first = int(rng.integers(1, its))
    with j.TiledSolver(planes, cs.weight, cs.pweights, its, devices=[0] * nb, cuts=cuts) as t:
        how = t.exchange()
        r1 = t.run(first, log=cs.log)
        r2 = t.run(its - first, log=cs.log)
        got = [t.download(c) for c in range(len(planes))]
    same = all(g.shape == r.fdata.shape and np.array_equal(g.view(np.uint32), r.fdata.view(np.uint32)) for g, r in zip(got, ref))
    if cs.log and same:
        rows = np.concatenate([r1, r2])
        same = np.allclose(rows, want_rows, rtol=1e-9, atol=1e-12)
    ran += 1
    used[how + ("" if ex == "direct" else f", norm {norm}")] = used.get(how + ("" if ex == "direct" else f", norm {norm}"), 0) + 1
    bad += not same
    if not same:
        print("DIFF " + cs.describe() + f"  cuts {cuts} {how} norm {norm} first {first}", flush=True)
print(f"tiled sweep seed {seed}: {ran} cases, {bad} differ from the whole-canvas solver; exchanges used: {used}")
sys.exit(1 if bad else 0)
