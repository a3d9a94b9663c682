#!/bin/bash
set -u
O=gpurun_out/r03h; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_baseline_configs_gpu.py tests/test_batch_gpu.py tests/test_debug_build_gpu.py tests/test_cli.py -m gpu -x -q -k "not config2 and not full_size" ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log
timeout 300 python tools/division_exhaustive.py all | tee $O/division_exhaustive.jsonl
for i in 1 2; do
( timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs ) 2>&1 | grep '^{' | tail -1 > $O/bench_n1_$i.json
python - <<PY
import json
d=json.load(open("$O/bench_n1_$i.json")); r=d["roofline"]
print("n1", d["value"], r["iteration_ms"], r["frac"], {k:v["avg_launch_ms"] for k,v in r["per_kernel"].items()})
PY
done
