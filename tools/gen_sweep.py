"""k_gradient duration vs number of wavefronts (3968-wide plane = 32 strips; 16 rows per wavefront):
is the kernel latency-bound per wavefront (time steps with the number of 4096-wave generations) or throughput-bound?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg2png_amd as j
from jpeg2png_amd import synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 3968
for H in (256, 512, 1024, 1536, 2048, 2560, 3072, 4096, 5120, 6144, 8192):
    planes = synth.make_planes(W, H, "444", 10, seed=3, y_only=True)
    s = j.Solver(planes, 0.3, [0.001], 40)
    s.enable_timing(2)
    for _ in range(2):
        s.reset(); s.run(40); s.sync()
    g, p, n = s.kernel_times()
    waves = ((W - 4 + 123) // 124) * (H // 16)
    print(f"{W}x{H}: {waves:6d} waves = {waves / 4096:5.2f} generations: k_gradient {g * 1e3:6.1f} us ({g * 1e9 / (W * H):.2f} ps/px)  k_project {p * 1e3:6.1f} us ({p * 1e9 / (W * H):.2f} ps/px)", flush=True)
    s.close()
