#!/usr/bin/env python3
"""configs[0] (512x512 4:2:0 joint -i 50) and configs[1] (1080p 4:4:4 -s -i 100) as bare loops, for
`rocprofv3 --kernel-trace --stats`: where do small planes spend their time — inside kernels or between them?"""
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg2png_amd as j
from jpeg2png_amd import synth

which = sys.argv[1] if len(sys.argv) > 1 else "0"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if which == "0":
    planes = synth.make_planes(512, 512, "420", 10, seed=1235)
    solvers = [j.Solver(planes, 0.3, [0.001] * 3, 50)]
    its, px = 50, 512 * 512 * 3
else:
    planes = synth.make_planes(1920, 1080, "444", 10, seed=1236)
    solvers = [j.Solver([p], 0.3 if c == 0 else 0.0, [0.001], 100) for c, p in enumerate(planes)]
    its, px = 100, 1920 * 1080 * 3
nip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
for fold in ((1, 0) if len(sys.argv) <= 3 else (int(sys.argv[3]),)):
    for s in solvers:
        if fold == 9:               # the library's own defaults
            continue
        s.debug_option(j.J2P_OPT_NORM_IN_PROJECT, nip if fold else 0)
        s.debug_option(j.J2P_OPT_NORM_FOLD, fold)
    t0 = None
    for r in range(reps + 1):
        if r == 1:
            t0 = time.perf_counter()
        for s in solvers:
            s.reset()
        for _ in range(its // 10):
            for s in solvers:
                s.run(10)
        for s in solvers:
            s.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f"config {which} fold {fold} nip {nip if fold else 0}: {dt * 1e3:.3f} ms per solve, {dt / its * 1e6:.1f} us per iteration, {px * its / dt / 1e9:.1f} Gpx-it/s", flush=True)
