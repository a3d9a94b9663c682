#!/bin/bash
# parity of the new code paths (16-byte row loads in k_project; norm tree inside k_project), then their A/B
set -u
O=gpurun_out/r02i
R=$GRAFT_REPO_ROOT
mkdir -p $R/$O
export TMPDIR=/tmp
python tools/ab_parity.py /tmp/p_base.npy > /dev/null 2>&1
python tools/ab_parity.py /tmp/p_nip.npy fold=1 nip=1 > /dev/null 2>&1
python tools/ab_parity.py /tmp/p_fold.npy fold=1 > /dev/null 2>&1
J2P_LIBRARY=$R/variants/libj2p_row.so python tools/ab_parity.py /tmp/p_row.npy > /dev/null 2>&1
J2P_LIBRARY=$R/variants/libj2p_row.so python tools/ab_parity.py /tmp/p_row_nip.npy fold=1 nip=1 rev=1 > /dev/null 2>&1
J2P_LIBRARY=$R/variants/libj2p_c.so python tools/ab_parity.py /tmp/p_c.npy > /dev/null 2>&1
python tools/ab_parity.py --cmp /tmp/p_base.npy /tmp/p_nip.npy /tmp/p_fold.npy /tmp/p_row.npy /tmp/p_row_nip.npy /tmp/p_c.npy | tee $O/parity.log
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
  run ntg_fold ntg "--norm-fold 1"
  run ntg_nip ntg "--norm-fold 1 --norm-in-project 1"
  run c_nip c "--norm-fold 1 --norm-in-project 1"
  run row row ""
  run row_nip row "--norm-fold 1 --norm-in-project 1"
done
