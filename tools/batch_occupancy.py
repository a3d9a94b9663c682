#!/usr/bin/env python3
"""How busy is the GPU while the batch engine runs (review r5 item 4: would ONE launch for several images pay)?
Input: the kernel trace of `rocprofv3 --kernel-trace --output-format csv -- python bench.py --config batch ...`.
Over the window of the phase kernels: the time at least one kernel was running (union of the intervals), the sum of
kernel durations (= average number of kernels in flight x busy time), idle gaps, and per kernel the launches, the
mean duration and the share of the summed time.  A batched launch removes launches and their boundaries; it cannot
win more than the idle share plus what the boundaries cost while other images' kernels are NOT there to fill them.
usage: python tools/batch_occupancy.py kernel_trace.csv [images]"""
import collections
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
images = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ev = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
ev.sort()
phase = [e for e in ev if e[2].startswith(("j2p::k_gradient", "j2p::k_project", "j2p::k_norm"))]
t0, t1 = phase[0][0], max(e[1] for e in phase)
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
total = sum(e - s for s, e, _ in win)
# time with exactly k kernels in flight
points = sorted([(s, 1) for s, _, _ in win] + [(e, -1) for _, e, _ in win])
depth, last, at = 0, t0, collections.Counter()
for t, d in points:
    at[depth] += t - last
    depth += d
    last = t
per = collections.defaultdict(lambda: [0, 0])
for s, e, n in win:
    per[n][0] += 1
    per[n][1] += e - s
wall = t1 - t0
out = {
    "window_ms": round(wall / 1e6, 2), "images": images or None,
    "gpu_busy_share": round(busy / wall, 4), "idle_ms": round((wall - busy) / 1e6, 2),
    "mean_kernels_in_flight_while_busy": round(total / busy, 2),
    "share_of_window_with_k_kernels_in_flight": {str(k): round(v / wall, 4) for k, v in sorted(at.items()) if v / wall >= 0.001},
    "launches_per_image": round(len(win) / images, 1) if images else None,
    "per_kernel": {n: {"launches": c, "mean_us": round(t / c / 1e3, 2), "share_of_summed_time": round(t / total, 4)}
                   for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:8]},
}
print(json.dumps(out))
