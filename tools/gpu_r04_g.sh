#!/bin/bash
set -u
O=gpurun_out/r04g
mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_tiled_c_gpu.py -m gpu -x -q --timeout 120 ) 2>&1 | tail -3
( timeout 200 python tools/sweep_tiled.py 60 51 ) 2>&1 | tail -2
( J2P_BENCH_ONE_DEVICE=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --size 4096 --no-cpu-baseline ) > $O/bench_2ranks.log 2>&1; echo "2-rank bench rc=$?"; grep '^{' $O/bench_2ranks.log | tail -1 > $O/bench_2ranks.json
python - <<PY
import json
d=json.load(open("$O/bench_2ranks.json"))
print(d["value"], d["config"]["engine"])
for o in d["other_configs"]:
    print("   ", o["config"][:60], o.get("Mpx_it_per_s"), o.get("bits_equal_to_the_whole_canvas_solve"), o.get("error"))
PY
