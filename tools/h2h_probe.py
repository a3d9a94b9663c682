#!/usr/bin/env python3
"""host-to-host probe: j2p_compute() on the 4096^2 -i 500 plane, n calls, the j2p_compute_timing() split of each.
--torch: import torch and touch the device first; --resident: a resident solve + download into numpy first (what bench.py
has done by the time it gets to its host_to_host leg)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
flags = [a for a in sys.argv[1:] if a.startswith("--")]
args = [a for a in sys.argv[1:] if not a.startswith("--")]
if "--torch" in flags:
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
import jpeg2png_amd as j  # noqa: E402
from jpeg2png_amd import synth  # noqa: E402

n = int(args[0]) if args else 5
planes = synth.make_planes(4096, 4096, "444", 10, seed=1237, y_only=True)
if "--resident" in flags or "--resident-nodl" in flags or "--resident-small" in flags:
    rp = synth.make_planes(2048, 2048, "444", 10, seed=1237, y_only=True) if "--resident-small" in flags else planes
    s = j.Solver(rp, 0.3, [0.001], 500)
    for _ in range(3):
        s.reset()
        s.run(500)
        s.sync()
    if "--resident-nodl" not in flags:
        keep = s.download(0)
    s.close()
planes[0].fdata = j.decode_plane(planes[0])
splits = []
_, secs = j.compute_c(planes, 0.3, [0.001], 500, repeat=n, splits=splits)
for s_, sp in zip(secs, splits):
    print(json.dumps({"library": os.path.basename(j.LIB_PATH), "context": " ".join(flags) or "plain", "ms": round(s_ * 1e3, 2), **{k: round(v, 2) for k, v in sp.items()}}), flush=True)
