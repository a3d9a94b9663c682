import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg2png_amd as j
from jpeg2png_amd import synth
sub = sys.argv[1] if len(sys.argv) > 1 else "444"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
planes = synth.make_planes(N, N, sub, 10, seed=5)
its = 50
s = j.Solver(planes, 0.3, [0.001] * 3, its)
s.enable_timing(2)
for _ in range(2):
    s.reset(); s.run(its); s.sync()
g, p, n = s.kernel_times()
px = s.W * s.H * 3
print(f"joint {sub} {N}: k_gradient<3> {g*1e3:.1f} us  k_project {p*1e3:.1f} us  -> {px/ (g+p) / 1e6:.1f} Gpx-ch-it/s kernel-only, samples {n}")
s1 = j.Solver(planes[:1], 0.3, [0.001], its)
s1.enable_timing(2)
for _ in range(2):
    s1.reset(); s1.run(its); s1.sync()
g, p, n = s1.kernel_times()
print(f"luma only: k_gradient<1> {g*1e3:.1f} us  k_project {p*1e3:.1f} us -> {s1.W*s1.H/(g+p)/1e6:.1f}")
