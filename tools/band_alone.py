#!/usr/bin/env python3
"""What the cross-band schedule of the C row tiling costs a band that has a GPU to itself — the figure that decides
the multi-GPU efficiency, measurable on ONE GPU: a 2048-row band of the 16384-wide plane next to a 48-row band (which
is idle almost all the time), against the same rows solved whole.  J2P_TILED_EXCHANGE=direct|copy (and
J2P_TILED_NORM=all with copy) select the schedule; J2P_BANDS_PER_GPU=k cuts the 2048 rows into k bands on streams of their
own (what a GPU holding k bands of the canvas would run: while one band waits for an event the others compute)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg2png_amd as j            # noqa: E402
from jpeg2png_amd import synth      # noqa: E402

W, its = 16384, 100
p = synth.make_planes(W, 2096, "444", 10, seed=1238, y_only=True)[0]


def timed(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


k = int(os.environ.get("J2P_BANDS_PER_GPU", "1"))
res = {"norm": os.environ.get("J2P_TILED_NORM", "default"), "bands_per_gpu": k}
cuts = [2048 * i // k // 16 * 16 for i in range(k)] + [2048, 2096]
with j.TiledSolver([p], 0.3, [0.001], its, devices=[0] * (k + 1), cuts=cuts) as t:
    res["exchange"] = t.exchange()
    def run():
        t.reset()
        t.run(its)
        t.sync()
    res["band_2048_next_to_band_48_us_per_iteration"] = round(timed(run) / its * 1e6, 2)
    res["band_threads_host_cpu_s"] = round(t.host_cpu_seconds(), 3)
with j.Solver([p], 0.3, [0.001], its) as s:
    def run():
        s.reset()
        s.run(its)
        s.sync()
    res["whole_2096_rows_us_per_iteration"] = round(timed(run) / its * 1e6, 2)
res["efficiency"] = round(res["whole_2096_rows_us_per_iteration"] / res["band_2048_next_to_band_48_us_per_iteration"], 4)
print(json.dumps(res))
