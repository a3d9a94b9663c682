#!/bin/bash
set -u
O=gpurun_out/r02e
R=$GRAFT_REPO_ROOT
mkdir -p $R/$O
export TMPDIR=/tmp
python tools/small_planes.py 0 40 0 2>&1 | tee $O/small0.log
python tools/small_planes.py 1 10 0 2>&1 | tee $O/small1.log
( timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -8 $O/pytest_gpu.log
