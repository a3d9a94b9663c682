#!/usr/bin/env python3
"""One jointly optimised image (all channels in one solver), resident: us per iteration and the digest of the planes it leaves.
    python tools/joint_probe.py W H SUBSAMPLING [ITERATIONS] [QUALITY] [TAG]
(J2P_LIBRARY picks the build; the digest tells same-bits from timing-only variants)"""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg2png_amd as j  # noqa: E402
from jpeg2png_amd import synth  # noqa: E402

W, H, sub = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
its = int(sys.argv[4]) if len(sys.argv) > 4 else 100
q = int(sys.argv[5]) if len(sys.argv) > 5 else 50
tag = sys.argv[6] if len(sys.argv) > 6 else os.path.basename(os.environ.get("J2P_LIBRARY", "release"))
planes = synth.make_planes(W, H, sub, q, seed=1238)
s = j.Solver(planes, 0.3, [0.001] * 3, its)


def run():
    s.reset()
    s.run(its)
    s.sync()


run()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    run()
    best = min(best, time.perf_counter() - t0)
h = hashlib.blake2b(digest_size=8)
for c in range(3):
    h.update(s.download(c).tobytes())
cw, ch = s.W, s.H
print(json.dumps({"image": f"{W}x{H} {sub} Q{q} joint -i {its}", "variant": tag, "us_per_iteration": round(best / its * 1e6, 2),
                  "G_channel_pixel_iterations_per_s": round(3 * cw * ch * its / best / 1e9, 1), "digest": h.hexdigest()}))
s.close()
