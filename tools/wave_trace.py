#!/usr/bin/env python3
"""Where a launch of the phase kernels spends its time, wavefront by wavefront (needs the -DJ2P_TRACE build):
    python tools/build_variant.py trace -DJ2P_TRACE
    J2P_LIBRARY=ab/libj2p_trace.so python tools/wave_trace.py W H [sub] [y|rgb] [iterations]
For each of the traced launches: span of the launch (first wavefront start -> last wavefront end), when the
wavefronts start, how long the first rows / blocks take to arrive, how long a wavefront lives, how many wavefronts
each CU got.  10 ns resolution (the constant 100 MHz clock)."""
import collections
import json
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import jpeg2png_amd as j            # noqa: E402
from jpeg2png_amd import synth      # noqa: E402

W, H = int(sys.argv[1]), int(sys.argv[2])
sub = sys.argv[3] if len(sys.argv) > 3 else "444"
y_only = (sys.argv[4] if len(sys.argv) > 4 else "y") == "y"
its = int(sys.argv[5]) if len(sys.argv) > 5 else 12
planes = synth.make_planes(W, H, sub, 10, seed=5, y_only=y_only)
n = len(planes)
s = j.Solver(planes, 0.3, [0.001] * n, 50)
s.run(its)            # warm
s.sync()
s.trace(True)
s.run(its)
s.sync()
rec = s.trace(False, fetch=True)
rec = rec[rec[:, 0] != 0]           # wavefronts past the launch's last strip leave their slot empty
s.close()
tag = (rec[:, 3] >> np.uint64(56)).astype(int)
seq = ((rec[:, 3] >> np.uint64(32)) & np.uint64(0xffffff)).astype(int)
hw = (rec[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
# HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13]; xcc in [27:24]
cu_key = ((hw >> 24) & 0xf) * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xf)
simd = (hw >> 4) & 3
out = {"plane": f"{W}x{H} {sub} {'Y' if y_only else 'joint'}", "launches": {}}
names = {1: "k_gradient", 2: "k_project"}
seqs = sorted(set(seq))[2:-1]                 # skip the first and the last traced iterations
for k in (1, 2):
    spans, starts, firsts, lives, percu, waves, gaps = [], [], [], [], [], [], []
    for q in seqs:
        m = (tag == k) & (seq == q)
        if not m.any():
            continue
        t0, t1, t2 = rec[m, 0].astype(np.int64), rec[m, 1].astype(np.int64), rec[m, 2].astype(np.int64)
        base = t0.min()
        spans.append((t2.max() - base) * 0.01)
        starts.append(np.percentile(t0 - base, [50, 90, 100]) * 0.01)
        ok = t1 > 0
        firsts.append(np.percentile((t1 - t0)[ok], [10, 50, 90]) * 0.01 if ok.any() else np.zeros(3))
        lives.append(np.percentile(t2 - t0, [10, 50, 90, 100]) * 0.01)
        c = collections.Counter(cu_key[m].tolist())
        percu.append((len(c), min(c.values()), max(c.values())))
        waves.append(int(m.sum()))
        # gap to the previous kernel of the stream: its last end -> this one's first start
        prev = (tag == (2 if k == 1 else 1)) & (seq == (q - 1 if k == 1 else q))
        if prev.any():
            gaps.append((base - rec[prev, 2].astype(np.int64).max()) * 0.01)
    if not spans:
        continue
    out["launches"][names[k]] = {
        "wavefronts": int(np.median(waves)),
        "span_us": round(float(np.median(spans)), 2),
        "gap_after_previous_kernel_us": round(float(np.median(gaps)), 2) if gaps else None,
        "wave_start_us_p50_p90_max": [round(float(v), 2) for v in np.median(np.array(starts), axis=0)],
        "first_data_after_start_us_p10_p50_p90": [round(float(v), 2) for v in np.median(np.array(firsts), axis=0)],
        "wave_life_us_p10_p50_p90_max": [round(float(v), 2) for v in np.median(np.array(lives), axis=0)],
        "cus_used_min_max_waves_per_cu": [int(v) for v in np.median(np.array(percu), axis=0)],
    }
print(json.dumps(out))
