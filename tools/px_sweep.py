#!/usr/bin/env python3
"""Geometry of the gradient strips against canvas size: columns per lane (J2P_PX) x rows per strip (J2P_RPW) on Y-only
and joint canvases from 0.26 to 16.8 Mpixel, solver-resident (reset + run), us per iteration.  One JSON line per case."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# (the switches this tool turns live in the experiments build of the library)
if "J2P_LIBRARY" not in os.environ:
    from jpeg2png_amd.buildlib import build_experiments
    os.environ["J2P_LIBRARY"] = build_experiments()
import jpeg2png_amd as j            # noqa: E402
from jpeg2png_amd import synth      # noqa: E402

CASES = [("512x512 420 joint", 512, 512, "420", False, 50), ("1024x1024 Y", 1024, 1024, "444", True, 50),
         ("1920x1080 Y", 1920, 1080, "444", True, 100), ("1920x1080 420 joint", 1920, 1080, "420", False, 50),
         ("2048x2048 Y", 2048, 2048, "444", True, 100), ("3072x2048 Y", 3072, 2048, "444", True, 100),
         ("4096x2048 Y", 4096, 2048, "444", True, 100), ("4096x4096 Y", 4096, 4096, "444", True, 100)]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if any(a in c[0] for a in sys.argv[1:])]
for name, W, H, sub, yonly, its in CASES:
    planes = synth.make_planes(W, H, sub, 10, seed=7, y_only=yonly)
    n = len(planes)
    row = {"case": name}
    for px in ("2", "1"):
        for rpw in ("16", "8", "4"):
            os.environ["J2P_PX"], os.environ["J2P_RPW"] = px, rpw
            s = j.Solver(planes, 0.3, [0.001] * n, its)

            def run():
                s.reset()
                s.run(its)
                s.sync()
            run()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                run()
            dt = (time.perf_counter() - t0) / reps
            s.close()
            row[f"px{px}_rpw{rpw}"] = round(dt / its * 1e6, 2)
    del os.environ["J2P_PX"], os.environ["J2P_RPW"]
    s = j.Solver(planes, 0.3, [0.001] * n, its)
    run()
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    row["policy"] = round((time.perf_counter() - t0) / 5 / its * 1e6, 2)
    s.close()
    print(json.dumps(row), flush=True)
