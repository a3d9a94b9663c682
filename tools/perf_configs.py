#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations on one GPU (not the bench line):
  configs[0] 512x512 4:2:0 Q10 -i 50 joint           (the reference's own CPU-runnable case)
  configs[1] 1920x1080 4:4:4 Q10 -i 100, Y/Cb/Cr as three compute(1,...) calls on three streams
  configs[4] slice: 8 x 1080p 4:2:0 Q50 -i 100 joint, images round-robin over streams
Prints one JSON line per configuration (Mpixel-iterations/s, canvas pixels x channels x iterations)."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import jpeg2png_amd as j  # noqa: E402
from jpeg2png_amd import synth  # noqa: E402


def timed(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def config0():
    planes = synth.make_planes(512, 512, "420", 10, seed=1235)
    s = j.Solver(planes, 0.3, [0.001] * 3, 50)

    def run():
        s.reset()
        s.run(50)
        s.sync()
    dt = timed(run, 10)
    return {"config": "512x512 4:2:0 Q10 -i 50 joint", "ms_per_solve": dt * 1e3,
            "Mpx_it_per_s": 512 * 512 * 3 * 50 / dt / 1e6}


def config1():
    planes = synth.make_planes(1920, 1080, "444", 10, seed=1236)
    solvers = [j.Solver([p], 0.3 if c == 0 else 0.0, [0.001], 100) for c, p in enumerate(planes)]

    def run():
        for s in solvers:
            s.reset()
        th = [threading.Thread(target=lambda s=s: (s.run(100), s.sync())) for s in solvers]
        for t in th:
            t.start()
        for t in th:
            t.join()
    dt = timed(run, 5)
    return {"config": "1920x1080 4:4:4 Q10 -i 100, 3 components on 3 streams (-s: chroma weight 0)",
            "ms_per_solve": dt * 1e3, "Mpx_it_per_s": 1920 * 1080 * 3 * 100 / dt / 1e6}


def config4(n=8):
    planes = synth.make_planes(1920, 1080, "420", 50, seed=1238)
    solvers = [j.Solver(planes, 0.3, [0.001] * 3, 100) for _ in range(n)]
    W, H = solvers[0].W, solvers[0].H

    def run():
        for s in solvers:
            s.reset()
        for s in solvers:
            s.run(100)
        for s in solvers:
            s.sync()
    dt = timed(run, 2)
    return {"config": f"{n} x 1080p 4:2:0 Q50 -i 100 joint (canvas {W}x{H}), one stream each",
            "ms_per_batch": dt * 1e3, "Mpx_it_per_s": n * W * H * 3 * 100 / dt / 1e6}


if __name__ == "__main__":
    j.build()
    for fn in (config0, config1, config4):
        print(json.dumps(fn()), flush=True)
