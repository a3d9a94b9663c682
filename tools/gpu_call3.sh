#!/bin/bash
set -u
O=gpurun_out/r02c
mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_batch_gpu.py -q ) > $O/pytest_batch.log 2>&1; echo "pytest batch rc=$?"; tail -3 $O/pytest_batch.log
python tools/small_planes.py 0 40 2>&1 | tee $O/small0.log
python tools/small_planes.py 1 10 2>&1 | tee $O/small1.log
cd /tmp
for w in 0 1; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_small$w -- python $GRAFT_REPO_ROOT/tools/small_planes.py $w 10 > /dev/null 2>&1
  f=$(find /tmp/prof_small$w -name '*kernel_stats.csv' | head -1)
  cp "$f" $GRAFT_REPO_ROOT/$O/small${w}_kernel_stats.csv
  head -12 "$f" | cut -c1-200
done
cd $GRAFT_REPO_ROOT
( timeout 300 python bench.py --config batch --steps 2 --warmup 1 --batch 32 ) > $O/bench_batch.log 2>&1; tail -c 800 $O/bench_batch.log
