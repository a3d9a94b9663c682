#!/bin/bash
set -u
O=gpurun_out/r02k
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_debug_build_gpu.py -x -q ) > $O/pytest_debug.log 2>&1; echo "pytest debug rc=$?"; tail -15 $O/pytest_debug.log
python tools/sweep_bands.py 60 3 2>&1 | grep -v "^ok\|amdgpu.ids" | cut -c1-300
python tools/sweep_bands.py 60 3 --split 2>&1 | grep -v "^ok\|amdgpu.ids" | cut -c1-300
python tools/sweep_vs_ref.py 120 31 2>&1 | tail -3
python tools/small_planes.py 0 40 9 2>&1 | grep config
python tools/small_planes.py 1 10 9 2>&1 | grep config
