#!/usr/bin/env python3
"""Per-launch durations of the phase kernels from a rocprofv3 kernel trace (csv): percentiles, by iteration parity, by
position in the solve, and the slowest launches — is the spread of k_gradient (StdDev 7 % of the mean in the kernel
stats) a few outliers, a drift, or two populations?
usage: python tools/launch_durations.py kernel_trace.csv     (of `bench.py --steps K`: solves of 500 iterations)"""
import csv
import json
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
out = {}
for key in ("k_gradient", "k_project", "k_norm_whole"):
    r = sorted(((int(x["Start_Timestamp"]), int(x["End_Timestamp"])) for x in rows if ("j2p::" + key) in x["Kernel_Name"]))
    if not r:
        continue
    d = np.array([(e - s) / 1e3 for s, e in r])
    idx = np.arange(len(d))
    out[key] = {
        "launches": len(d), "mean_us": round(float(d.mean()), 2), "std_us": round(float(d.std()), 2),
        "p1_p10_p50_p90_p99_max": [round(float(v), 2) for v in np.percentile(d, [1, 10, 50, 90, 99, 100])],
        "mean_even_odd_launches": [round(float(d[idx % 2 == 0].mean()), 2), round(float(d[idx % 2 == 1].mean()), 2)],
        "mean_by_tenth_of_the_run": [round(float(v.mean()), 2) for v in np.array_split(d, 10)],
        # bench.py's solves are 500 iterations each: position inside the solve (the first iterations take the IEEE rows of
        # the short division more often: flat areas are rounding noise around 0 until the iterate has moved)
        "mean_by_iteration_in_solve_0_10_20_30_40_50_100_200": [round(float(d[(idx % 500 >= a_) & (idx % 500 < b_)].mean()), 2)
                                                                 for a_, b_ in ((0, 10), (10, 20), (20, 30), (30, 40), (40, 50), (50, 100), (100, 200), (200, 500))],
        "share_above_1p15_x_median": round(float((d > 1.15 * np.median(d)).mean()), 4),
        "mean_without_those": round(float(d[d <= 1.15 * np.median(d)].mean()), 2),
    }
print(json.dumps(out))
