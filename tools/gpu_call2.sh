#!/bin/bash
set -u
O=gpurun_out/r02b
mkdir -p $O
export TMPDIR=/tmp
for fold in 1 0 1 0; do
  ( timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --norm-fold $fold ) >> $O/bench_ab.log 2>&1
done
grep -o '"value": [0-9.]*\|"norm_fold": [a-z]*\|"avg_launch_ms": [0-9.]*' $O/bench_ab.log | paste - - - -
( timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -15 $O/pytest_gpu.log
( timeout 600 python bench.py --steps 3 --warmup 1 ) > $O/bench_full.log 2>&1; echo "bench full rc=$?"; tail -c 2500 $O/bench_full.log
