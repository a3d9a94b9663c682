#!/usr/bin/env python3
"""Non-temporal policy by the DEVICE's live working sets (default) against by the solver's own (J2P_NT_SCOPE=solver, the
round-2 policy), where several solvers share a GPU: eight 1080p 4:2:0 joint images on eight streams (the configs[4]
slice of bench.py, ~100 MB each) and eight 256-row bands of a 16384-wide plane (72 MiB each).  ms per batch / per solve."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import time
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    out = {"scope": os.environ.get("J2P_NT_SCOPE", "device")}
    planes = synth.make_planes(1920, 1080, "420", 50, seed=1238)
    solvers = [j.Solver(planes, 0.3, [0.001] * 3, 100) for _ in range(8)]

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
    for _ in range(4):
        run()
    out["eight_1080p_420_images_ms"] = round((time.perf_counter() - t0) / 4 * 1e3, 3)
    for s in solvers:
        s.close()
    p = synth.make_planes(16384, 2048, "444", 10, seed=1238, y_only=True)
    with j.TiledSolver(p, 0.3, [0.001], 100, devices=[0] * 8) as t:
        def run2():
            t.reset()
            t.run(100)
            t.sync()
        run2()
        t0 = time.perf_counter()
        for _ in range(3):
            run2()
        out["eight_256_row_bands_of_16384x2048_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
    print(json.dumps(out))
    sys.exit(0)
for scope in ("device", "solver", "device", "solver"):
    env = dict(os.environ)
    if scope == "solver":
        env["J2P_NT_SCOPE"] = "solver"
    else:
        env.pop("J2P_NT_SCOPE", None)
    r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
    print((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
