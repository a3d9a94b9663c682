import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg2png_amd as j
from jpeg2png_amd import synth
for (W, H, sub, yonly) in ((512, 512, "420", False), (1920, 1080, "420", False), (512, 512, "444", True)):
    planes = synth.make_planes(W, H, sub, 10, seed=5, y_only=yonly)
    for its in (50, 100, 200):
        s = j.Solver(planes, 0.3, [0.001] * len(planes), its)
        for rep in range(3):
            s.reset(); s.sync()
            t0 = time.perf_counter(); s.run(its); t1 = time.perf_counter(); s.sync(); dt = time.perf_counter() - t0
            print(f"{W}x{H} {sub} its={its} rep{rep}: submit {1e6*(t1-t0)/its:.1f} us/it, wall {dt/its*1e6:.1f} us/it")
        s.close()
