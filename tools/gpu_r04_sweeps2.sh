#!/bin/bash
# more randomised parity sweeps on the round's final library, further seeds
set -u
export TMPDIR=/tmp
( timeout 900 python tools/sweep_vs_ref.py 1500 61 ) 2>&1 | tail -1
( timeout 400 python tools/sweep_tiled.py 300 62 ) 2>&1 | tail -1
( timeout 400 python tools/sweep_bands.py 150 63 ) 2>&1 | tail -1
( timeout 400 python tools/sweep_bands.py 80 64 --split ) 2>&1 | tail -1
( python -c "from jpeg2png_amd.buildlib import build_debug; build_debug()" && J2P_LIBRARY=jpeg2png_amd/libjpeg2png_amd_debug.so timeout 400 python tools/debug_sweep.py 60 65 ) 2>&1 | tail -1
