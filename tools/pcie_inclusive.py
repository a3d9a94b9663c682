#!/usr/bin/env python3
"""Whole-call rate of the drop-in path at the headline size: host buffers in, host planes out
(upload + device decode + aux_init + 500 iterations + download), i.e. what compute() costs a
caller that starts and ends in host memory.  Prints Mpixel-iterations/s."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg2png_amd as j  # noqa: E402
from jpeg2png_amd import synth  # noqa: E402

W = H = 4096
its = 500
planes = synth.make_planes(W, H, "444", 10, seed=1237, y_only=True)
planes[0].fdata = j.decode_plane(planes[0])
for rep in range(3):
    import copy
    p = copy.deepcopy(planes)
    t0 = time.perf_counter()
    j.compute(p, 0.3, [0.001], its)
    dt = time.perf_counter() - t0
    print(f"compute() host-to-host: {dt*1e3:.1f} ms  -> {W*H*its/dt/1e6:.0f} Mpx-it/s")
