#!/bin/bash
# the fold without acknowledgement waits (parity in the sign bit of the partials): parity of every norm schedule, then timing
set -u
O=gpurun_out/r04l
mkdir -p $O
export TMPDIR=/tmp
( timeout 500 python -m pytest tests/test_parity_gpu.py tests/test_tiled_c_gpu.py -m gpu -x -q --timeout 300 -k "schedule or tiled or band or fold or norm or compute_matches or drop_in or exchange or cuts" ) 2>&1 | grep -E "passed|failed|error|Error" | tail -3
for rep in 1 2; do
for mode in "-1 -1" "1 0" "1 2"; do
  set -- $mode
  ( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-host-to-host --norm-fold $1 --norm-in-project $2 ) 2>&1 | grep '^{' | tail -1 > $O/tmp.json
  python - <<PY
import json
d=json.load(open("$O/tmp.json")); r=d["roofline"]
print(json.dumps({"headline 4096^2 -i 500, norm_fold":$1,"norm_in_project":$2,"Mpx_it_per_s":d["value"],"us_per_iteration":round(r["iteration_ms"]*1e3,2),"k_gradient_us":round(r["per_kernel"]["k_gradient"]["avg_launch_ms"]*1e3,2),"k_project_us":round(r["per_kernel"]["k_project"]["avg_launch_ms"]*1e3,2)}))
PY
done
done | tee $O/fold_noack_headline.jsonl
for sz in "1920 1080" "2048 2048" "4096 2048" "16384 2048" "8192 8192"; do
  for mode in "-1 -1" "1 0" "1 2"; do
    set -- $sz $mode
    ( timeout 200 python bench.py --size $1 --height $2 --iterations 100 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-to-host --norm-fold $3 --norm-in-project $4 ) 2>&1 | grep '^{' | tail -1 > $O/tmp.json
    python - <<PY
import json
d=json.load(open("$O/tmp.json")); r=d["roofline"]
print(json.dumps({"plane":"$1x$2 Y-only Q10 -i 100","norm_fold":$3,"norm_in_project":$4,"Mpx_it_per_s":d["value"],"us_per_iteration":round(r["iteration_ms"]*1e3,2)}))
PY
  done
done | tee $O/fold_noack_sizes.jsonl
for w in all counter; do J2P_TILED_WAIT=$w J2P_TILED_EXCHANGE=direct timeout 300 python tools/band_alone.py; done 2>&1 | grep '^{' | tee $O/band_alone.jsonl
