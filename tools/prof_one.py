import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg2png_amd as j
from jpeg2png_amd import synth
which = sys.argv[1]
if which == "c0":
    planes = synth.make_planes(512, 512, "420", 10, seed=1235); its = 50
else:
    planes = synth.make_planes(1920, 1080, "420", 50, seed=1238); its = 100
s = j.Solver(planes, 0.3, [0.001] * 3, its)
for _ in range(3):
    s.reset(); s.run(its); s.sync()
