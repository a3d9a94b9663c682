#!/usr/bin/env python3
"""A slice of the randomised case stream (tests/sweep_cases.py) through whatever library J2P_LIBRARY names —
whole canvas and the C row tiling with three bands on one GPU — printing a digest of every resulting plane and,
when the library is the checked build (-DJ2P_DEBUG), the number of address violations its phase kernels counted.
    J2P_LIBRARY=jpeg2png_amd/libjpeg2png_amd_debug.so python tools/debug_sweep.py [ncases] [seed]
Exit status 1 on any violation.  tests/test_debug_build_gpu.py runs it once per build and compares the digests."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import jpeg2png_amd as j  # noqa: E402
from jpeg2png_amd import tiled  # noqa: E402
from sweep_cases import cases  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4
checked = j.debug_build()
digest = hashlib.sha1()
violations = 0


def tally(solver, what):
    global violations
    if not checked:
        return
    count, site, off = solver.debug_violations()
    if count:
        violations += count
        print(f"   {what}: {count} address violations, first at site {site}, offset {off}")


extra = [(4096, 96, "444", True), (1920, 1080, "420", False)]       # a wide plane with every fast path; the padded-canvas case
for cs in list(cases(seed, n)):
    planes = cs.planes()
    its = min(cs.iterations, 6)
    with j.Solver(planes, cs.weight, cs.pweights, its) as s:
        s.run(its, log=cs.log)
        for c in range(len(planes)):
            digest.update(s.download(c).tobytes())
        tally(s, "whole canvas")
    align = tiled.band_alignment(planes)
    H = max(p.h * p.h_samp for p in planes)
    if (H + align - 1) // align >= 3:
        with j.TiledSolver(planes, cs.weight, cs.pweights, its, devices=[0, 0, 0]) as t:
            t.run(its, log=cs.log)
            for c in range(len(planes)):
                digest.update(t.download(c).tobytes())
            for b in range(3):
                tally(t.band_solver(b), f"band {b}")
    print("case", cs.describe(), flush=True)
from jpeg2png_amd import synth  # noqa: E402
for (w, h, sub, yonly) in extra:
    planes = synth.make_planes(w, h, sub, 10, seed=5, y_only=yonly)
    with j.Solver(planes, 0.3, [0.001] * len(planes), 4) as s:
        s.run(4)
        for c in range(len(planes)):
            digest.update(s.download(c).tobytes())
        tally(s, f"{w}x{h} {sub}")
print(f"checked build: {checked}; violations: {violations}; digest {digest.hexdigest()}")
sys.exit(1 if violations else 0)
