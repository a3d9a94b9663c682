#!/bin/bash
# A/B on one box: projection walking order (Infinity Cache), non-temporal g / x; then where small planes spend their time
set -u
O=gpurun_out/r02g
R=$GRAFT_REPO_ROOT
mkdir -p $R/$O
export TMPDIR=/tmp
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs"
run() {  # name, library, extra args
  local lib=""
  [ -n "$2" ] && lib="J2P_LIBRARY=$R/variants/libj2p_$2.so"
  env $lib $B $3 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel']
print('$1', d['value'], 'us/it', round(d['ms_per_step']*1000/500,2), 'grad', r['k_gradient']['avg_launch_ms'], 'proj', r['k_project']['avg_launch_ms'])" | tee -a $O/ab.log
}
for rep in 1 2; do
  run base "" ""
  run reverse "" "--proj-reverse 1"
  run ntg ntg ""
  run ntg_rev ntg "--proj-reverse 1"
  run ntgx ntgx ""
  run ntgx_rev ntgx "--proj-reverse 1"
done
python tools/small_planes.py 0 40 0 2>&1 | tee $O/small0.log
python tools/small_planes.py 0 40 1 2>&1 | tee -a $O/small0.log
python tools/small_planes.py 1 10 0 2>&1 | tee $O/small1.log
cd /tmp
for w in 0 1; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_small$w -- python $R/tools/small_planes.py $w 10 0 > $R/$O/rocprof_small$w.log 2>&1
  f=$(find /tmp/prof_small$w -name '*kernel_stats.csv' | head -1)
  cp "$f" $R/$O/small${w}_kernel_stats.csv
  head -8 "$f" | cut -c1-160
  t=$(find /tmp/prof_small$w -name '*kernel_trace.csv' | head -1)
  python3 - "$t" <<'PY'
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-400:]
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
print("last 400 dispatches: median kernel %.2f us, median gap %.2f us, mean gap %.2f us, span per dispatch %.2f us" % (
    statistics.median(durs) / 1e3, statistics.median(gaps) / 1e3, sum(gaps) / len(gaps) / 1e3,
    (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / len(rows) / 1e3))
for r in rows[-8:]:
    print(r["Kernel_Name"][:50], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
done
