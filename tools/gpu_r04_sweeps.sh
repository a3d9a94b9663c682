#!/bin/bash
# randomised parity sweeps on the round's final library, new seeds
set -u
O=gpurun_out/r04_sweeps
mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python tools/sweep_tiled.py 300 41 ) 2>&1 | tail -3
( timeout 900 python tools/sweep_vs_ref.py 800 42 ) 2>&1 | tail -2
( timeout 600 python tools/sweep_wide.py 300 43 ) 2>&1 | tail -2
( timeout 600 python tools/sweep_cli.py 80 44 ) 2>&1 | tail -2
( timeout 300 python tools/sweep_odd_params.py 9 ) 2>&1 | tail -2
( timeout 300 python tools/two_channel.py ) 2>&1 | tail -2
