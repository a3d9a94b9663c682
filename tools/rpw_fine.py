#!/usr/bin/env python3
"""Rows per gradient strip at a finer grain than the policy's 4 / 8 / 16 (J2P_RPW = any value 2..64, whole canvases):
does the number of wavefronts per SIMD (strips x segments / 1024) explain what mid-size canvases lose?  One JSON line per
canvas: us per iteration and the k_gradient launch average (HIP events around every launch would serialise the stream,
so the kernel time is the solver's own phase log: j2p_solver_phase_ms) for every value."""
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

CASES = {"1920x1080": (1920, 1080, [4, 5, 6, 7, 8, 9, 10, 12, 16]),
         "2048x2048": (2048, 2048, [8, 10, 11, 12, 13, 14, 16, 20, 24]),
         "3072x2048": (3072, 2048, [8, 12, 14, 16, 18, 20, 24]),
         "4096x2048": (4096, 2048, [12, 16, 18, 20, 22, 23, 24, 28, 32]),
         "4096x4096": (4096, 4096, [16, 18, 20, 24, 32])}
names = sys.argv[1:] or list(CASES)
for name in names:
    W, H, values = CASES[name]
    planes = synth.make_planes(W, H, "444", 10, seed=7, y_only=True)
    its = 100
    row = {"case": name + " Y", "strips": (W - 4 + 123) // 124}
    for rpw in values:
        os.environ["J2P_RPW"] = str(rpw)
        s = j.Solver(planes, 0.3, [0.001], its)

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
        waves = row["strips"] * ((H + rpw - 1) // rpw)
        row[f"rpw{rpw}"] = {"us": round(dt / its * 1e6, 2), "waves_per_simd": round(waves / 1024, 2)}
    del os.environ["J2P_RPW"]
    print(json.dumps(row), flush=True)
