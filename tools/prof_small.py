import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg2png_amd as j
from jpeg2png_amd import synth
for (W, H, sub, yonly) in ((1920, 1080, "444", True), (1920, 1080, "420", False), (512, 512, "420", False), (512, 512, "444", True)):
    planes = synth.make_planes(W, H, sub, 10, seed=5, y_only=yonly)
    its = 100
    s = j.Solver(planes, 0.3, [0.001] * len(planes), its)
    s.enable_timing(1)
    for _ in range(2):
        s.reset(); s.run(its); s.sync()
    g, p, n = s.kernel_times()
    s.enable_timing(0)
    s.reset(); s.sync()
    t0 = time.perf_counter(); s.run(its); s.sync(); dt = time.perf_counter() - t0
    print(f"{W}x{H} {sub} nch={len(planes)}: k_gradient {g*1e3:.1f} us  k_project {p*1e3:.1f} us  per-iteration wall {dt/its*1e6:.1f} us")
