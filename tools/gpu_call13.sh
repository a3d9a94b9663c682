#!/bin/bash
# k_project workgroup-to-strip mappings and load order, same box A/B; batch bench
set -u
O=gpurun_out/r02m
R=$GRAFT_REPO_ROOT
mkdir -p $R/$O
export TMPDIR=/tmp
for v in pvert pxcd porder; do J2P_LIBRARY=$R/variants/libj2p_$v.so python tools/ab_parity.py /tmp/p_$v.npy > /dev/null 2>&1; done
python tools/ab_parity.py /tmp/p_base.npy > /dev/null 2>&1
python tools/ab_parity.py --cmp /tmp/p_base.npy /tmp/p_pvert.npy /tmp/p_pxcd.npy /tmp/p_porder.npy | tee $O/parity.log
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs"
run() {  # name, library, extra args
  local lib=""
  [ -n "$2" ] && lib="J2P_LIBRARY=$R/variants/libj2p_$2.so"
  env $lib $B $3 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel']
print('$1', d['value'], 'us/it', round(d['roofline']['iteration_ms']*1000,2), 'grad', r['k_gradient']['avg_launch_ms'], 'proj', r['k_project']['avg_launch_ms'])" | tee -a $O/ab.log
}
for rep in 1 2; do
  run base "" ""
  run pvert pvert ""
  run pxcd pxcd ""
  run porder porder ""
done
( timeout 300 python bench.py --config batch --steps 2 --warmup 1 --batch 32 ) 2>&1 | grep '^{' | tee $O/bench_batch.json | cut -c1-1200
