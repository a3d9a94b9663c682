#!/usr/bin/env python3
"""configs[4] through the C batch engine, host to host: output arrays allocated per job (what bench.py did) against a
ring of pre-touched arrays reused between jobs — does mapping host memory while other jobs' kernels run cost throughput?"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import jpeg2png_amd as j  # noqa: E402
from jpeg2png_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 8
planes = synth.make_planes(1920, 1080, "420", 50, seed=1238)
for mode in ("fresh", "ring", "fresh", "ring"):
    ring = [np.zeros((1080, 1920, 3), np.uint8) for _ in range(4 * slots)] if mode == "ring" else None
    with j.Batch(devices=[0], slots_per_device=slots) as b:
        def step(k):
            pending = []
            for i in range(k):
                if ring is not None and len(pending) >= len(ring):
                    b.wait(pending.pop(0))
                pending.append(b.submit(planes, 0.3, [0.001] * 3, 100, width=1920, height=1080, bits=8,
                                        out=None if ring is None else ring[i % len(ring)]))
            for t in pending:
                b.wait(t)
        step(4 * slots)
        t0 = time.perf_counter()
        step(n)
        dt = time.perf_counter() - t0
    print(json.dumps({"outputs": mode, "images": n, "slots": slots, "images_per_s": round(n / dt, 1)}), flush=True)
