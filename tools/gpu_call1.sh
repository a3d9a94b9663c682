#!/bin/bash
# first GPU call of round 2: smoke, the new test files, A/B of the folded norm reduction
set -u
O=gpurun_out/r02a
mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python __graft_entry__.py --smoke ) > $O/smoke.log 2>&1; echo "smoke rc=$?"
( timeout 600 python -m pytest tests/test_tiled_c_gpu.py tests/test_batch_gpu.py -x -q ) > $O/pytest_new.log 2>&1; echo "pytest new rc=$?"; tail -5 $O/pytest_new.log
( timeout 900 python -m pytest tests/test_baseline_configs_gpu.py -x -q -k "not config2" ) > $O/pytest_configs.log 2>&1; echo "pytest configs rc=$?"; tail -5 $O/pytest_configs.log
( timeout 600 python -m pytest tests/test_cli.py -x -q -m gpu ) > $O/pytest_cli.log 2>&1; echo "pytest cli rc=$?"; tail -5 $O/pytest_cli.log
for fold in 1 0 1 0; do
  ( timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --norm-fold $fold ) >> $O/bench_ab.log 2>&1
done
grep -o '"value": [0-9.]*\|"norm_fold": [a-z]*\|"avg_launch_ms": [0-9.]*' $O/bench_ab.log | paste - - - - 
( timeout 600 python bench.py --steps 3 --warmup 1 ) > $O/bench_full.log 2>&1; echo "bench full rc=$?"; tail -c 3000 $O/bench_full.log
