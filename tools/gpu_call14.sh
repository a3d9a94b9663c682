#!/bin/bash
set -u
O=gpurun_out/r02n
R=$GRAFT_REPO_ROOT
mkdir -p $R/$O
export TMPDIR=/tmp
python tools/ab_parity.py /tmp/p_base.npy > /dev/null 2>&1
python tools/ab_parity.py /tmp/p_lead.npy lead=1 > /dev/null 2>&1
python tools/ab_parity.py --cmp /tmp/p_base.npy /tmp/p_lead.npy | tee $O/parity.log
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs"
run() {  # name, library, extra args
  $B $3 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel']
print('$1', d['value'], 'us/it', round(d['roofline']['iteration_ms']*1000,2), 'grad', r['k_gradient']['avg_launch_ms'], 'proj', r['k_project']['avg_launch_ms'])" | tee -a $O/ab.log
}
for rep in 1 2 3; do
  run base "" ""
  run lead "" "--norm-leader 1"
done
python tools/small_planes.py 0 40 9 2>&1 | grep config
python tools/small_planes.py 0 40 7 2>&1 | grep config
python tools/small_planes.py 1 10 9 2>&1 | grep config
python tools/small_planes.py 1 10 7 2>&1 | grep config
