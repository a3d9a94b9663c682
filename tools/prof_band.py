"""timing aid: one row band (rows 0..2048 of a 16384 x 4096 Y plane, the per-GPU share of the N=2 bench
workload) stepped without neighbours — for rocprofv3 --kernel-trace --stats (k_rowsums / k_norm_finish cost)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg2png_amd as j
from jpeg2png_amd import synth

W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
planes = synth.make_planes(W, 4096, "444", 10, seed=3, y_only=True)
s = j.Solver(planes, 0.3, [0.001], 50, band=(0, 2048))
for _ in range(50):
    s.phase_gradient()
    s.phase_project()
s.sync()
