#!/bin/bash
# small / mid canvases: the solver's own norm schedule (fold + per-wavefront tree in k_project) against the reducer workgroup
set -u
export TMPDIR=/tmp
for mode in -1 2 -1 2; do J2P_MODE=$mode timeout 120 python - <<PY
import json, os, sys, time
sys.path.insert(0, ".")
import jpeg2png_amd as j
from jpeg2png_amd import synth
mode = int(os.environ["J2P_MODE"])
out = {"norm_fold": mode}
for name, W, H, sub, its in (("512x512 420 joint", 512, 512, "420", 50), ("1024x1024 Y", 1024, 1024, "y", 50), ("1920x1080 420 joint", 1920, 1080, "420", 50), ("1920x1080 Y", 1920, 1080, "y", 50), ("1536x1536 Y", 1536, 1536, "y", 50), ("256x256 420 joint", 256, 256, "420", 50)):
    planes = synth.make_planes(W, H, "444" if sub == "y" else sub, 10, seed=1235, y_only=(sub == "y"))
    with j.Solver(planes, 0.3, [0.001] * len(planes), its) as s:
        if mode >= 0:
            s.debug_option(j.J2P_OPT_NORM_IN_PROJECT, 0)
            s.debug_option(j.J2P_OPT_NORM_FOLD, mode)
        def run():
            s.reset(); s.run(its); s.sync()
        for _ in range(5): run()
        t0 = time.perf_counter()
        for _ in range(30): run()
        out[name] = round((time.perf_counter() - t0) / 30 / its * 1e6, 2)
print(json.dumps(out))
PY
done 2>&1 | grep '^{'
