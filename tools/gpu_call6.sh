#!/bin/bash
# re-entry of round 2: whole GPU suite, then the rocprof evidence for the bench line
set -u
O=gpurun_out/r02f
mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --durations=15 ) > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -30 $O/pytest_gpu.log
bash tools/collect_profiles.sh r02 2>&1 | tail -12
