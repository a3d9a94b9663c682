#!/bin/bash
# round 3, call D: side-stream schedule of the row tiling (parity, then what it buys), in-projection norm tree
# behind the row loads (small planes)
set -u
O=gpurun_out/r03d
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_tiled_c_gpu.py tests/test_batch_gpu.py tests/test_debug_build_gpu.py tests/test_baseline_configs_gpu.py tests/test_parity_gpu.py -m gpu -x -q -k "not config2 and not vs_reference_i4" ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 $O/pytest_gpu.log
for side in 1 0; do J2P_TILED_SIDE=$side timeout 300 python tools/band_alone.py | tee -a $O/band_alone.jsonl; done
for side in 1 0; do
  ( J2P_TILED_SIDE=$side timeout 600 python bench.py --force-tiled --bands 8 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --tiled-impl c ) 2>&1 | grep '^{' | tail -1 > $O/bench_tiled_8bands_side$side.json
  python - <<PY
import json
d=json.load(open("$O/bench_tiled_8bands_side$side.json"))
print("8 bands on one GPU, side=$side:", d["value"], d["ms_per_step"], d["config"].get("band_threads_host_cpu_s"))
PY
done
( timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ) 2>&1 | grep '^{' | tail -1 > $O/bench_n1.json
python - <<PY
import json
d=json.load(open("$O/bench_n1.json")); r=d["roofline"]
print("n1", d["value"], r["iteration_ms"], r["frac"], {k:v["avg_launch_ms"] for k,v in r["per_kernel"].items()})
for o in d["other_configs"]: print("  ", o)
PY
