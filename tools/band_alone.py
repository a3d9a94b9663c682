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
# (the switches this tool turns live in the experiments build of the library)
if "J2P_LIBRARY" not in os.environ:
    from jpeg2png_amd.buildlib import build_experiments
    os.environ["J2P_LIBRARY"] = build_experiments()
import jpeg2png_amd as j            # noqa: E402
from jpeg2png_amd import synth      # noqa: E402

W, its = 16384, 100
# J2P_DUMMY_BANDS=d: d idle 48-row bands instead of one (the big band then waits for d gradient events per iteration, as
# on a (d + 1)-GPU node); J2P_BIG_BAND_LAST=1 puts them in front, so that the big band is not the copy exchange's root
dummies = int(os.environ.get("J2P_DUMMY_BANDS", "1"))
big_last = os.environ.get("J2P_BIG_BAND_LAST", "0") == "1"
H = 2048 + 48 * dummies
p = synth.make_planes(W, H, "444", 10, seed=1238, y_only=True)[0]


def timed(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


k = int(os.environ.get("J2P_BANDS_PER_GPU", "1"))
res = {"norm": os.environ.get("J2P_TILED_NORM", "default"), "bands_per_gpu": k, "dummy_bands": dummies, "big_band_last": big_last}
if big_last:
    cuts = [48 * i for i in range(dummies)] + [48 * dummies + 2048 * i // k // 16 * 16 for i in range(k)] + [H]
else:
    cuts = [2048 * i // k // 16 * 16 for i in range(k)] + [2048 + 48 * i for i in range(dummies)] + [H]
with j.TiledSolver([p], 0.3, [0.001], its, devices=[0] * (k + dummies), cuts=cuts) as t:
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
    res["whole_us_per_iteration"] = round(timed(run) / its * 1e6, 2)
res["rows_whole"] = H
res["efficiency"] = round(res["whole_us_per_iteration"] * 2048 / H / res["band_2048_next_to_band_48_us_per_iteration"], 4)
print(json.dumps(res))
