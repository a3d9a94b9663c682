#!/usr/bin/env python3
"""Same-box A/B of the iteration schedules on one-channel planes (us per iteration, resident: reset + run + sync):
   old        what round 4 ran: fold + per-wavefront tree up to 2.5 Mpixel (two launches), a k_norm_whole launch above (three)
   two        two launches, ||g|| folded into k_gradient and reduced by every k_project wavefront (J2P_OPT_FUSE 0)
   fused      ONE launch per iteration (k_iterate), every projection wavefront reduces ||g||
   fused_wg   ... the workgroup's first wavefront reduces ||g|| (J2P_OPT_NORM_IN_PROJECT 2)
usage: python tools/fuse_ab.py [iterations] [W H ...]"""
import json
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# (the switches this tool turns live in the experiments build of the library)
if "J2P_LIBRARY" not in os.environ:
    from jpeg2png_amd.buildlib import build_experiments
    os.environ["J2P_LIBRARY"] = build_experiments()
import jpeg2png_amd as j  # noqa: E402
from jpeg2png_amd import synth  # noqa: E402

only = None
if "--only" in sys.argv:
    k = sys.argv.index("--only")
    only = sys.argv[k + 1].split(",")
    del sys.argv[k:k + 2]
its = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sizes = [(512, 512), (1024, 1024), (1920, 1080), (2048, 2048), (4096, 2048), (4096, 4096)]
if len(sys.argv) > 3:
    v = [int(x) for x in sys.argv[2:]]
    sizes = list(zip(v[0::2], v[1::2]))


def timed(s, reps=5):
    def run():
        s.reset()
        s.run(its)
        s.sync()
    run()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        run()
        best = min(best, time.perf_counter() - t0)
    return best / its * 1e6


for W, H in sizes:
    planes = synth.make_planes(W, H, "444", 10, seed=1234 + 3, y_only=True)
    out = {"plane": f"{W}x{H} Y-only Q10 -i {its}", "library": os.path.basename(j.LIB_PATH)}
    digests = {}
    for name, opts in (("old", "old"), ("two", {j.J2P_OPT_FUSE: 0}), ("fused", {j.J2P_OPT_FUSE: 1}),
                       ("fused_wg", {j.J2P_OPT_FUSE: 1, j.J2P_OPT_NORM_IN_PROJECT: 2})):
        if only and name not in only:
            continue
        with j.Solver(planes, 0.3, [0.001], its) as s:
            if opts == "old":
                s.debug_option(j.J2P_OPT_FUSE, 0)
                if W * H > 5 << 19:
                    s.debug_option(j.J2P_OPT_NORM_IN_PROJECT, 0)
                    s.debug_option(j.J2P_OPT_NORM_FOLD, 0 if W * H < 1 << 25 else 1)
            else:
                for k, v in opts.items():
                    s.debug_option(k, v)
            out[name + "_us"] = round(timed(s), 2)
            out[name + "_launches"] = s.launches_per_iteration()
            import hashlib
            digests[name] = hashlib.blake2b(s.download(0), digest_size=8).hexdigest()
    out["same_bits"] = len(set(digests.values())) == 1
    print(json.dumps(out), flush=True)
