#!/bin/bash
# the norm reduction as the last workgroup of the gradient launch (J2P_OPT_NORM_FOLD = 2): parity, then timing against the
# k_norm_whole launch
set -u
O=gpurun_out/r04l
mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -x -q --timeout 120 -k "schedule_switch" ) 2>&1 | grep -E "passed|failed|error|Error|Timeout" | tail -3
for rep in 1 2; do
for mode in "-1" "2"; do
  ( timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-host-to-host --norm-fold $mode ) 2>&1 | grep '^{' | tail -1 > $O/tmp.json
  python - <<PY
import json
d=json.load(open("$O/tmp.json")); r=d["roofline"]
print(json.dumps({"headline 4096^2 -i 500, norm_fold":$mode,"Mpx_it_per_s":d["value"],"us_per_iteration":round(r["iteration_ms"]*1e3,2),"k_gradient_us":round(r["per_kernel"]["k_gradient"]["avg_launch_ms"]*1e3,2),"k_project_us":round(r["per_kernel"]["k_project"]["avg_launch_ms"]*1e3,2)}))
PY
done
done | tee $O/reducer_headline.jsonl
for sz in "1920 1080" "2048 2048" "4096 2048" "4096 5120" "8192 4096"; do
  for mode in "-1" "2"; do
    set -- $sz $mode
    ( timeout 120 python bench.py --size $1 --height $2 --iterations 100 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-to-host --norm-fold $3 ) 2>&1 | grep '^{' | tail -1 > $O/tmp.json
    python - <<PY
import json
d=json.load(open("$O/tmp.json")); r=d["roofline"]
print(json.dumps({"plane":"$1x$2 Y-only Q10 -i 100","norm_fold":$3,"Mpx_it_per_s":d["value"],"us_per_iteration":round(r["iteration_ms"]*1e3,2)}))
PY
  done
done | tee $O/reducer_sizes.jsonl
