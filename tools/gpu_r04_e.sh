#!/bin/bash
set -u
O=gpurun_out/r04e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ex in direct copy; do
  rm -rf /tmp/tr_$ex
  ( J2P_TILED_EXCHANGE=$ex timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$ex -- python $R/tools/band_alone.py ) > $R/$O/rocprof_$ex.log 2>&1
  python $R/tools/band_trace.py /tmp/tr_$ex > $R/$O/band_trace_$ex.txt 2>&1; head -40 $R/$O/band_trace_$ex.txt
  find /tmp/tr_$ex -name "*kernel_trace.csv" -exec cp {} $R/$O/kernel_trace_$ex.csv \;
done
