#!/usr/bin/env python3
"""Where a band solver finishes ||g||: inside k_project (the workgroup's first wavefront runs the tree over the global
row sums, NIP 2; J2P_BAND_NIP=1, default) against a k_norm_finish launch between the phases (J2P_BAND_NIP=0).  One band
— rows [0, 2048) of a 16384 x 4096 canvas, 1024 global tile rows, the shape of configs[3] — driven with the phase
calls on one GPU; the global array is whatever the arena holds (timing only)."""
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

W, H, rows, its = 16384, 4096, 2048, 100
p = synth.make_planes(W, H, "444", 10, seed=1238, y_only=True)[0]
out = {}
for nip in ("1", "0", "1", "0"):
    os.environ["J2P_BAND_NIP"] = nip
    with j.Solver([p], 0.3, [0.001], its, band=(0, rows)) as s:
        def run():
            for _ in range(its):
                s.phase_gradient()
                s.phase_project()
            s.sync()
        run()
        t0 = time.perf_counter()
        for _ in range(3):
            run()
        out.setdefault("nip_" + nip, []).append(round((time.perf_counter() - t0) / 3 / its * 1e6, 2))
print(json.dumps({"band": f"{W}x{rows} of a {H}-row canvas", "us_per_iteration": out}))
