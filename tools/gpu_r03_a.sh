#!/bin/bash
# round 3, call A: the whole GPU suite on the reworked tiling / batch / CLI code, the bench line, the round-1
# decompositions of the phase kernels repeated on the CURRENT kernels, the multi-band figures
set -u
O=gpurun_out/r03a
mkdir -p $O
export TMPDIR=/tmp
( timeout 1700 python -m pytest tests -m gpu -x -q --durations=12 ) > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -40 $O/pytest_gpu.log
( timeout 600 python bench.py --steps 5 --warmup 2 ) > $O/bench_n1.log 2>&1; echo "bench rc=$?"; grep '^{' $O/bench_n1.log | tail -1 > $O/bench_n1.json; cut -c1-1800 $O/bench_n1.json
for v in noarith notraffic nohalo; do
  ( J2P_LIBRARY=variants/libj2p_$v.so timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs ) 2>&1 | grep '^{' | tail -1 > $O/bench_$v.json
  python - <<PY
import json
d=json.load(open("$O/bench_$v.json")); r=d["roofline"]
print("$v", d["value"], r["iteration_ms"], {k:v["avg_launch_ms"] for k,v in r["per_kernel"].items()})
PY
done
( timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs ) 2>&1 | grep '^{' | tail -1 > $O/bench_base.json
python - <<PY
import json
d=json.load(open("$O/bench_base.json")); r=d["roofline"]
print("base", d["value"], r["iteration_ms"], {k:v["avg_launch_ms"] for k,v in r["per_kernel"].items()})
PY
# two ranks on this box's one GPU (gloo): the driver's --gpus N launch shape, rank 0 driving two bands
( J2P_BENCH_ONE_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --size 4096 ) > $O/bench_2ranks.log 2>&1; echo "2-rank bench rc=$?"; grep '^{' $O/bench_2ranks.log | tail -1 > $O/bench_2ranks_1gpu_gloo.json; cut -c1-1500 $O/bench_2ranks_1gpu_gloo.json
# 8 bands of 16384x2048 on one GPU, both norm schedules
for m in root all; do
  ( J2P_TILED_NORM=$m timeout 600 python bench.py --force-tiled --bands 8 --steps 2 --warmup 1 --no-cpu-baseline --tiled-impl c ) 2>&1 | grep '^{' | tail -1 > $O/bench_tiled_8bands_$m.json
  python - <<PY
import json
d=json.load(open("$O/bench_tiled_8bands_$m.json"))
print("8 bands, norm=$m:", d["value"], d["ms_per_step"], d["config"].get("band_threads_host_cpu_s"), [ (o.get("config","")[:40], o.get("Mpx_it_per_s"), o.get("images_per_s")) for o in d.get("other_configs",[])])
PY
done
