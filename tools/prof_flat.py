"""how much do flat grey areas (chroma noise around 0 -> IEEE fallback rows) cost?  1080p 4:2:0 Q10 joint,
and the same with the flat-area fraction raised by zeroing chroma coefficient blocks."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jpeg2png_amd as j
from jpeg2png_amd import synth

its = 100
for frac in (None, 0.0, 0.5, 1.0):
    planes = synth.make_planes(1920, 1080, "420", 10, seed=4321)
    zb = []
    for p in planes[1:]:
        d = p.data.reshape(p.h // 8, p.w // 8, 64)
        if frac is not None:
            if frac == 0.0:
                d[(np.abs(d).sum(axis=2) == 0)] = 1          # no all-zero block anywhere (DC = 1)
            else:
                d[: int(d.shape[0] * frac)] = 0
        zb.append(float((np.abs(d).sum(axis=2) == 0).mean()))
    s = j.Solver(planes, 0.3, [0.001] * 3, its)
    s.enable_timing(2)
    for _ in range(2):
        s.reset(); s.run(its); s.sync()
    g, p, n = s.kernel_times()
    print(f"zero chroma blocks {zb[0]:.2f}: k_gradient {g*1e3:.1f} us  k_project {p*1e3:.1f} us", flush=True)
