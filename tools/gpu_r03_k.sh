#!/bin/bash
set -u
O=gpurun_out/r03k; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "schedule_switch or fixed or sweep or selftest or both_joint or noise" ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log
timeout 600 python tools/px_sweep.py | tee $O/px_rpw_sweep.jsonl
( J2P_LIBRARY=variants/libj2p_noslp.so timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs ) 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('noslp', d['value'], r['iteration_ms'], {k:v['avg_launch_ms'] for k,v in r['per_kernel'].items()})"
( timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs ) 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('base', d['value'], r['iteration_ms'], {k:v['avg_launch_ms'] for k,v in r['per_kernel'].items()})"
