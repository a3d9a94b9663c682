#!/bin/bash
# the round's evidence in one call: whole GPU suite, the multi-rank bench control flow on one GPU, profiles
set -u
TAG=${1:-r02}
O=gpurun_out/${TAG}_final
mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -14 $O/pytest_gpu.log
( timeout 300 python __graft_entry__.py --smoke ) 2>&1 | tail -1
# two ranks on this box's one GPU (gloo instead of RCCL): the C row tiling driven by rank 0, as the driver's --gpus N run does
( J2P_BENCH_ONE_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --size 4096 ) > $O/bench_2ranks.log 2>&1; echo "2-rank bench rc=$?"; grep '^{' $O/bench_2ranks.log | cut -c1-700
( timeout 300 python bench.py --force-tiled --bands 4 --steps 2 --warmup 1 ) 2>&1 | grep '^{' | tee $O/bench_tiled_4bands_1gpu.json | cut -c1-600
( timeout 300 python bench.py --config batch --steps 2 --warmup 1 --batch 32 ) 2>&1 | grep '^{' | tee $O/bench_batch.json | cut -c1-600
bash tools/collect_profiles.sh $TAG 2>&1 | tail -12
