#!/usr/bin/env python3
"""rows per gradient strip against concurrency: one 1080p 4:2:0 joint image alone, and eight of them on eight
streams (the configs[4] slice of bench.py), with 16- and 8-row strips; ms per image / per batch."""
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

planes = synth.make_planes(1920, 1080, "420", 50, seed=1238)
out = {}
for rpw in ("16", "8", ""):
    if rpw:
        os.environ["J2P_RPW"] = rpw
    else:
        os.environ.pop("J2P_RPW", None)
    for n in (1, 3, 8):
        solvers = [j.Solver(planes, 0.3, [0.001] * 3, 100) for _ in range(n)]

        def run():
            for s in solvers:
                s.reset()
            for _ in range(10):
                for s in solvers:
                    s.run(10)
            for s in solvers:
                s.sync()
        run()
        t0 = time.perf_counter()
        for _ in range(3):
            run()
        dt = (time.perf_counter() - t0) / 3
        for s in solvers:
            s.close()
        out[f"rpw{rpw or 'policy'}_x{n}_ms"] = round(dt * 1e3, 3)
print(json.dumps(out))
