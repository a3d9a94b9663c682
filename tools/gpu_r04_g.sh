#!/bin/bash
set -u
O=gpurun_out/r04g
mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_tiled_c_gpu.py -m gpu -x -q ) 2>&1 | tail -2
for w in all root collector; do J2P_TILED_WAIT=$w J2P_TILED_EXCHANGE=direct timeout 300 python tools/band_alone.py; done 2>&1 | grep '^{' | tee $O/band_alone_wait_modes.jsonl
( timeout 900 python bench.py --force-tiled --bands 8 --steps 2 --warmup 1 --no-cpu-baseline ) 2>&1 | grep '^{' | tail -1 > $O/bench_tiled_8bands_1gpu.json
python - <<PY
import json
d=json.load(open("$O/bench_tiled_8bands_1gpu.json"))
print(d["value"], d["config"]["engine"])
for o in d["other_configs"]:
    print("   ", o["config"][:60], o.get("Mpx_it_per_s"), o.get("bits_equal_to_the_whole_canvas_solve"), o.get("error"))
PY
