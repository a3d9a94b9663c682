#!/bin/bash
set -u
O=gpurun_out/r02j
R=$GRAFT_REPO_ROOT
mkdir -p $R/$O
export TMPDIR=/tmp
J2P_LIBRARY=$R/variants/libj2p_ntg.so python tools/ab_parity.py /tmp/p_old.npy > /dev/null 2>&1
python tools/ab_parity.py /tmp/p_new.npy > /dev/null 2>&1
python tools/ab_parity.py /tmp/p_new_nip.npy fold=1 nip=1 > /dev/null 2>&1
python tools/ab_parity.py --cmp /tmp/p_old.npy /tmp/p_new.npy /tmp/p_new_nip.npy | tee $O/parity.log
python tools/small_planes.py 0 40 0 2>&1 | grep config | tee $O/small0.log
python tools/small_planes.py 0 40 1 1 2>&1 | grep config | tee -a $O/small0.log
python tools/small_planes.py 0 40 1 0 2>&1 | grep config | tee -a $O/small0.log
python tools/small_planes.py 1 10 0 2>&1 | grep config | tee $O/small1.log
python tools/small_planes.py 1 10 1 1 2>&1 | grep config | tee -a $O/small1.log
( timeout 1200 python -m pytest tests -m gpu -q -x -k "not config2" ) > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -5 $O/pytest_gpu.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' | tee $O/bench.json | cut -c1-1500
