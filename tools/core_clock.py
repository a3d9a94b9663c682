#!/usr/bin/env python3
"""The shader clock the phase kernels actually run at (needs the -DJ2P_TRACE -DJ2P_TRACE_CLOCK build):
    python tools/build_variant.py traceclk -DJ2P_TRACE -DJ2P_TRACE_CLOCK
    J2P_LIBRARY=ab/libj2p_traceclk.so python tools/core_clock.py [W H]
Every wavefront records its life twice: on the constant 100 MHz clock (s_memrealtime) and in core-clock ticks (s_memtime);
the ratio is the clock the SIMDs ran at while the kernel was executing — what the cycle counts of tools/isa_count.py
and tools/ubench/valu_rates have to be priced at.  One JSON line."""
import json
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import jpeg2png_amd as j            # noqa: E402
from jpeg2png_amd import synth      # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 4096)
planes = synth.make_planes(W, H, "444", 10, seed=5, y_only=True)
s = j.Solver(planes, 0.3, [0.001], 100)
s.run(60)            # warm: the clock has settled under load
s.sync()
s.trace(True)
s.run(8)
s.sync()
rec = s.trace(False, fetch=True)
rec = rec[rec[:, 0] != 0]
s.close()
tag = (rec[:, 3] >> np.uint64(56)).astype(int)
out = {"plane": f"{W}x{H} Y"}
for k, name in ((1, "k_gradient"), (2, "k_project")):
    m = tag == k
    wall = (rec[m, 2] - rec[m, 0]).astype(np.float64) * 10.0      # ns
    core = rec[m, 1].astype(np.float64)
    ok = wall > 2000
    ghz = core[ok] / wall[ok]
    out[name] = {"wavefronts": int(ok.sum()), "core_clock_ghz_p10_p50_p90": [round(float(v), 3) for v in np.percentile(ghz, [10, 50, 90])],
                 "wave_life_us_p50": round(float(np.median(wall[ok])) * 1e-3, 2), "core_ticks_per_wave_p50": int(np.median(core[ok]))}
print(json.dumps(out))
