#!/usr/bin/env python3
"""Timeline of the row tiling's launches from a rocprofv3 kernel trace (csv): two iterations from the middle of the
run, µs relative to the first launch shown, per queue — where a band's iteration goes when it has a GPU (almost) to
itself.   rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/band_alone.py ; python tools/band_trace.py DIR"""
import csv
import glob
import os
import sys

root = sys.argv[1]
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_gradient", "k_project", "k_norm", "k_copy_rows"))]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# band_alone.py runs the tiled solver first, then the same rows whole (k_norm_whole appears only there)
whole_from = min([r["s"] for r in rows if "k_norm_whole" in r["Kernel_Name"]] or [1 << 62])
rows = [r for r in rows if r["s"] < whole_from]
# the tiled run comes first in band_alone.py: big-band gradient launches = the largest grid among k_gradient launches
grads = [r for r in rows if "k_gradient" in r["Kernel_Name"]]
import collections
sizes = collections.Counter(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) for r in grads)
big = max(k for k, v in sizes.items() if v >= 100)
queues = {}
for r in rows:
    queues.setdefault(r["Queue_Id"], 0)
    queues[r["Queue_Id"]] += 1
tiled = [r for r in rows if len(queues) > 1]
biggrads = [r for r in grads if int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) == big]
# iterations of the tiled run: take launches 150..152 of the big band's gradient
q_big = biggrads[150]["Queue_Id"]
t0 = biggrads[150]["s"]
t1 = biggrads[152]["s"]
print(f"{len(rows)} launches; two iterations of the tiled run, us from the big band's gradient launch:")
for r in rows:
    if t0 <= r["s"] < t1:
        name = r["Kernel_Name"].replace("void j2p::", "").replace("j2p::", "")[:48]
        print(f"  {(r['s'] - t0) / 1e3:9.2f} {(r['e'] - t0) / 1e3:9.2f} dur {(r['e'] - r['s']) / 1e3:7.2f}  q{r['Queue_Id']}  grid {int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])}  {name}")
per = (biggrads[350]["s"] - biggrads[150]["s"]) / 200 / 1e3
busy = 0
for r in rows:
    if r["Queue_Id"] == q_big and biggrads[150]["s"] <= r["s"] < biggrads[350]["s"]:
        busy += r["e"] - r["s"]
print(f"per iteration {per:.1f} us; the big band's queue has a kernel running {busy / 200 / 1e3:.1f} us of it")
