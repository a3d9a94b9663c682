"""kernel times of the headline plane (and a 12 Mpixel joint image) under the segment-length knob J2P_RPW
(read at solver creation) — run once per J2P_GRAD_WPB value, which is read once per process."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg2png_amd as j
from jpeg2png_amd import synth

which = sys.argv[1] if len(sys.argv) > 1 else "y"
if which == "y":
    planes = synth.make_planes(4096, 4096, "444", 10, seed=1237, y_only=True)
else:
    planes = synth.make_planes(3500, 3500, "420", 10, seed=5)
its = 60
for rpw in (16, 32, 48, 64, 128):
    os.environ["J2P_RPW"] = str(rpw)
    s = j.Solver(planes, 0.3, [0.001] * len(planes), its)
    s.enable_timing(2)
    for _ in range(2):
        s.reset(); s.run(its); s.sync()
    g, p, n = s.kernel_times()
    print(f"{which} WPB {os.environ.get('J2P_GRAD_WPB', '4')} RPW {rpw:3d}: k_gradient {g*1e3:6.1f} us  k_project {p*1e3:6.1f} us", flush=True)
    s.close()
