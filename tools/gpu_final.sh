#!/bin/bash
# the round's evidence in one call: whole GPU suite, profiles (kernel stats, counters, bench line), the multi-rank bench
# control flow on one GPU, the row tiling and the batch engine on one GPU, sweeps on the final kernels
set -u
TAG=${1:-r03}
O=gpurun_out/${TAG}_final
mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -26 $O/pytest_gpu.log
( timeout 300 python __graft_entry__.py --smoke ) 2>&1 | tail -1
bash tools/collect_profiles.sh $TAG 2>&1 | tail -6
# two ranks on this box's one GPU (gloo): the driver's --gpus N launch shape, rank 0 driving two bands, both C schedules
( J2P_BENCH_ONE_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --size 4096 ) > $O/bench_2ranks.log 2>&1; echo "2-rank bench rc=$?"; grep '^{' $O/bench_2ranks.log | tail -1 > gpurun_out/${TAG}_bench_2ranks_1gpu_gloo.json; cut -c1-400 gpurun_out/${TAG}_bench_2ranks_1gpu_gloo.json
# 8 bands of 16384x2048 on one GPU (C engine, both schedules; strong-scaling denominator; 256-image batch)
( timeout 600 python bench.py --force-tiled --bands 8 --steps 2 --warmup 1 --no-cpu-baseline ) 2>&1 | grep '^{' | tail -1 > gpurun_out/${TAG}_bench_tiled_8bands_1gpu.json; cut -c1-300 gpurun_out/${TAG}_bench_tiled_8bands_1gpu.json
# the RCCL harness with one rank as its own neighbour, plain and split schedule
for ov in 0 1; do
  ( J2P_TILED_SELF_NEIGHBOURS=1 J2P_RCCL_OVERLAP=$ov timeout 300 python bench.py --force-tiled --tiled-impl rccl --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs ) 2>&1 | grep '^{' | tail -1 > $O/rccl_self_$ov.json
  python - <<PY
import json
try:
    d=json.load(open("$O/rccl_self_$ov.json")); print(json.dumps({"rccl_harness_one_rank_self_neighbours": True, "overlap": $ov, "Mpx_it_per_s": d["value"], "us_per_iteration_of_one_2048_row_band": round(d["ms_per_step"]*1e3/100/2,2)}))
except Exception as e: print("rccl self leg failed", e)
PY
done | tee gpurun_out/${TAG}_rccl_self_neighbours.jsonl
( timeout 300 python bench.py --config batch --steps 2 --warmup 1 --batch 32 ) 2>&1 | grep '^{' | tee gpurun_out/${TAG}_bench_batch.json | cut -c1-300
for combo in "0 root" "0 all" "1 root"; do set -- $combo; J2P_TILED_SPLIT=$1 J2P_TILED_NORM=$2 timeout 300 python tools/band_alone.py; done | tee gpurun_out/${TAG}_band_alone.jsonl
timeout 300 python tools/nt_scope.py | tee gpurun_out/${TAG}_nt_scope.jsonl
# size sweep on the final kernels
for sz in "1920 1080" "2048 2048" "4096 2048" "4096 4096" "4096 5120" "8192 4096" "16384 2048" "8192 8192"; do
  set -- $sz
  ( timeout 200 python bench.py --size $1 --height $2 --iterations 100 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs ) 2>&1 | grep '^{' | tail -1 > $O/tmp.json
  python - <<PY
import json
d=json.load(open("$O/tmp.json")); r=d["roofline"]
print(json.dumps({"plane":"$1x$2 Y-only Q10 -i 100","Mpx_it_per_s":d["value"],"us_per_iteration":round(r["iteration_ms"]*1e3,2),"iteration_frac":r["frac"],"k_gradient_us":round(r["per_kernel"]["k_gradient"]["avg_launch_ms"]*1e3,1),"k_project_us":round(r["per_kernel"]["k_project"]["avg_launch_ms"]*1e3,1)}))
PY
done | tee gpurun_out/${TAG}_size_sweep.jsonl
for c in "512 512 420 rgb" "1920 1080 444 y" "2048 2048 444 y" "4096 4096 444 y"; do
  ( J2P_LIBRARY=ab/libj2p_trace.so timeout 120 python tools/wave_trace.py $c ) 2>&1 | grep '^{'
done | tee gpurun_out/${TAG}_wave_trace.jsonl
# the shader clock the kernels run at, and the cost of a device-wide barrier against a launch boundary
python tools/build_variant.py traceclk -DJ2P_TRACE -DJ2P_TRACE_CLOCK > /dev/null 2>&1
for sz in "4096 4096" "2048 2048"; do J2P_LIBRARY=ab/libj2p_traceclk.so timeout 200 python tools/core_clock.py $sz; done | tee gpurun_out/${TAG}_core_clock.jsonl
[ -x tools/ubench/grid_sync ] || hipcc --offload-arch=gfx950 -O3 -o tools/ubench/grid_sync tools/ubench/grid_sync.hip
timeout 120 tools/ubench/grid_sync 1000 | tee gpurun_out/${TAG}_grid_sync.json
