"""the bench workload itself against the UNMODIFIED reference: 4096x4096 Y-only Q10, -i 500, weight 0.3,
pweight 0.001 (BASELINE configs[2], same seed as bench.py).  ~90 s of one CPU core for the reference."""
import copy
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import jpeg2png_amd as j
from jpeg2png_amd import synth
from oracle import bindings as oracle

its = int(sys.argv[1]) if len(sys.argv) > 1 else 500
planes = synth.make_planes(4096, 4096, "444", 10, seed=1234 + 3, y_only=True)
for p in planes:
    p.fdata = j.decode_plane(p)
t0 = time.perf_counter()
want, _, secs = oracle.ref_compute(planes, 0.3, [0.001], its)
t1 = time.perf_counter()
got = copy.deepcopy(planes)
j.compute(got, 0.3, [0.001], its)
t2 = time.perf_counter()
same = np.array_equal(got[0].fdata.view(np.uint32), want[0].view(np.uint32))
mse = float(np.mean((got[0].fdata.astype(np.float64) - want[0].astype(np.float64)) ** 2))
psnr = float("inf") if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
print(f"4096x4096 Y Q10 -i {its}: bit-identical {same}, PSNR {psnr} dB; reference {secs:.1f} s in compute(), "
      f"GPU compute() host-to-host {t2 - t1:.3f} s")
sys.exit(0 if psnr >= 80 else 1)
