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
slot_all = np.arange(rec.shape[0])
keep = rec[:, 0] != 0               # wavefronts past the launch's last strip leave their slot empty
slot_all = slot_all[keep]
rec = rec[keep]
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
    # the per-SIMD picture of ONE launch (the middle one of those traced): how many wavefronts each SIMD got, how long it
    # was busy (sum of its wavefronts' lives / its residency), when it ran dry (busy-until), and how full the chip was
    q = seqs[len(seqs) // 2]
    m = (tag == k) & (seq == q)
    t0, t2 = rec[m, 0].astype(np.int64), rec[m, 2].astype(np.int64)
    base = t0.min()
    span = (t2.max() - base) * 0.01
    skey = cu_key[m] * 4 + simd[m]
    keys = np.unique(skey)
    nw = np.array([(skey == s_).sum() for s_ in keys])
    until = np.array([(t2[skey == s_].max() - base) * 0.01 for s_ in keys])
    first = np.array([(t0[skey == s_].min() - base) * 0.01 for s_ in keys])
    life = np.array([((t2 - t0)[skey == s_]).sum() * 0.01 for s_ in keys])
    # wavefronts resident on the chip over the launch, sampled every 0.5 us
    ts = np.arange(0, span, 0.5)
    resident = np.array([((t0 - base) * 0.01 <= t).sum() - ((t2 - base) * 0.01 <= t).sum() for t in ts])
    hist = collections.Counter(nw.tolist())
    out["launches"][names[k]]["per_simd"] = {
        "launch_seq": int(q), "span_us": round(float(span), 2), "simds_used": int(len(keys)),
        "wavefronts_per_simd_histogram": {str(a): int(b) for a, b in sorted(hist.items())},
        "busy_until_us_p10_p50_p90_max": [round(float(v), 2) for v in np.percentile(until, [10, 50, 90, 100])],
        "first_start_us_p50_p90_max": [round(float(v), 2) for v in np.percentile(first, [50, 90, 100])],
        "busy_until_us_by_wavefront_count": {str(a): round(float(np.median(until[nw == a])), 2) for a in sorted(hist)},
        "mean_resident_wavefronts_per_simd": round(float(life.sum() / (len(keys) * span)), 2),
        "resident_wavefronts_per_simd_over_time_every_5us": [round(float(v) / len(keys), 2) for v in resident[::10]],
    }
    if k == 1 and y_only and all(__import__("os").environ.get(v) == "0" for v in ("J2P_ZONE_D", "J2P_ZONE_B", "J2P_ZONE_C")):
        # who are the long-lived wavefronts?  (whole tile rows only: slot -> workgroup -> (tile row, strip), grad_item)
        sl = slot_all[m] - slot_all[m].min()
        b_, wv = sl // 4, sl % 4
        ntx = (W - 4 + 123) // 124
        ntr = (H + 15) // 16
        n = (ntx * ((ntr + 1) // 2) + 3) // 4       # units: four strips x a PAIR of tile rows, two workgroups each
        qx, jx = b_ & 7, b_ >> 3
        cb = qx * (n >> 3) + np.minimum(qx, n & 7)
        ident = 4 * (cb + (jx >> 1)) + wv
        trow, wcol = 2 * (ident // ntx) + (jx & 1), ident % ntx
        life_w = (t2 - t0) * 0.01
        start_w = (t0 - base) * 0.01
        xcc = (hw[m] >> 24) & 0xf
        def by(keyv, nbins=None):
            out_ = {}
            for v in sorted(set(keyv.tolist())):
                sel = keyv == v
                out_[str(v)] = [round(float(np.median(life_w[sel])), 2), round(float(np.percentile(life_w[sel], 95)), 2), int(sel.sum())]
            return out_
        long_ = life_w > np.percentile(life_w, 98)
        out["launches"][names[k]]["who_lives_long"] = {
            "life_p50_p95_count_by_xcd": by(xcc),
            "life_p50_p95_count_by_strip_column": by(wcol),
            "life_p50_p95_count_by_start_5us": by((start_w // 5).astype(int) * 5),
            "life_p50_p95_count_by_tile_row_mod_32": by(trow % 32),
            "life_p50_p95_count_by_simd": by(simd[m]),
            "wavefronts_by_simd_x_wave_in_workgroup": np.bincount(simd[m] * 4 + wv, minlength=16).reshape(4, 4).tolist(),
            "top2pct": {"n": int(long_.sum()), "strip_columns": collections.Counter(wcol[long_].tolist()).most_common(8),
                        "xcd": collections.Counter(xcc[long_].tolist()).most_common(8),
                        "tile_rows_mod_32": collections.Counter((trow[long_] % 32).tolist()).most_common(8),
                        "start_5us": collections.Counter(((start_w[long_] // 5).astype(int) * 5).tolist()).most_common(12)},
        }
print(json.dumps(out))
