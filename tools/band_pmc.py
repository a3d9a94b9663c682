#!/usr/bin/env python3
"""The workload behind profiles/r06_pmc_band.json: two 2048-row bands of the 16384-wide plane (BASELINE configs[3]'s
per-GPU shape) through the C row tiling on ONE GPU, exchange `direct` — the band kernels (k_gradient with the row-sum
push, k_project with NIP 2 and the halo push) as a GPU of an N-GPU run executes them.  Run under rocprofv3 --pmc passes:
    cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIR -- python tools/band_pmc.py
The counters come out per launch; with two equal bands every k_gradient / k_project launch IS one band's.
usage: python tools/band_pmc.py [W rows_per_band iterations]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("J2P_TILED_EXCHANGE", "direct")
import jpeg2png_amd as j            # noqa: E402
from jpeg2png_amd import synth      # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
its = int(sys.argv[3]) if len(sys.argv) > 3 else 20
plane = synth.make_y_plane_banded(W, 2 * rows, 10, seed=1238, band_rows=rows, workers=2)
with j.TiledSolver([plane], 0.3, [0.001], its, devices=[0, 0]) as t:
    t.run(its)
    t.sync()
    print("exchange:", t.exchange(), file=sys.stderr)
