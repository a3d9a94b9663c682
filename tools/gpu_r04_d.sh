#!/bin/bash
set -u
O=gpurun_out/r04d
mkdir -p $O
export TMPDIR=/tmp
for k in 1 2 3 4 1 2; do J2P_BANDS_PER_GPU=$k J2P_TILED_EXCHANGE=direct timeout 300 python tools/band_alone.py; done 2>&1 | grep '^{' | tee $O/band_alone_per_gpu.jsonl
for k in 1 2; do J2P_BANDS_PER_GPU=$k J2P_TILED_EXCHANGE=copy timeout 300 python tools/band_alone.py; done 2>&1 | grep '^{' | tee -a $O/band_alone_per_gpu.jsonl
