#!/bin/bash
set -u
O=gpurun_out/r02d
R=$GRAFT_REPO_ROOT
mkdir -p $R/$O
export TMPDIR=/tmp
cd /tmp
for w in 0 1; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_small$w -- python $R/tools/small_planes.py $w 10 0 > $R/$O/rocprof_small$w.log 2>&1
  f=$(find /tmp/prof_small$w -name '*kernel_stats.csv' | head -1)
  cp "$f" $R/$O/small${w}_kernel_stats.csv
  head -8 "$f" | cut -c1-220
  t=$(find /tmp/prof_small$w -name '*kernel_trace.csv' | head -1)
  # inter-kernel gaps on the busiest stream: start(i+1) - end(i) for the last 400 dispatches
  python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-400:]
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
import statistics
print("last 400 dispatches: median kernel %.2f us, median gap %.2f us, mean gap %.2f us, span per dispatch %.2f us" % (
    statistics.median(durs) / 1e3, statistics.median(gaps) / 1e3, sum(gaps) / len(gaps) / 1e3,
    (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / len(rows) / 1e3))
for r in rows[-8:]:
    print(r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us", "queue", r.get("Queue_Id"))
PY
done
