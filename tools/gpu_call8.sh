#!/bin/bash
# cache-policy variants on top of non-temporal g (same box A/B), and a footprint sweep (does Infinity-Cache residency pay?)
set -u
O=gpurun_out/r02h
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
for rep in 1 2; do
  run ntg ntg ""
  run a_pgld a ""
  run b_d b ""
  run c_xp c ""
  run d_st d ""
  run abc abc ""
  run abc_rev abc "--proj-reverse 1"
done
for h in 512 1024 2048 8192; do
  run base_h$h "" "--height $h --iterations 200"
  run ntg_h$h ntg "--height $h --iterations 200"
done
