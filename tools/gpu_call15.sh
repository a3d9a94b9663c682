#!/bin/bash
set -u
O=gpurun_out/r02p
R=$GRAFT_REPO_ROOT
mkdir -p $R/$O
export TMPDIR=/tmp
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs"
run() {  # name, library, extra args
  local lib=""
  [ -n "$2" ] && lib="J2P_LIBRARY=$R/variants/libj2p_$2.so"
  env $lib $B $3 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel']
print('$1', d['value'], 'us/it', round(d['roofline']['iteration_ms']*1000,2), 'grad', r['k_gradient']['avg_launch_ms'], 'proj', r['k_project']['avg_launch_ms'])" | tee -a $O/ab.log
}
for rep in 1 2 3; do
  run base "" ""
  run pw7 pw7 ""
  run pw8 pw8 ""
done
