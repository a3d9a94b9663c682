#!/bin/bash
set -u
export TMPDIR=/tmp
for rpw in 2 3 4 6 8; do J2P_RPW=$rpw timeout 120 python - <<PY
import json, os, sys, time
sys.path.insert(0, ".")
import jpeg2png_amd as j
from jpeg2png_amd import synth
out = {"J2P_RPW": int(os.environ["J2P_RPW"])}
for name, W, H, sub, its in (("512x512 420 joint -i 50", 512, 512, "420", 50), ("256x256 420 joint -i 50", 256, 256, "420", 50), ("1024x768 420 joint -i 50", 1024, 768, "420", 50), ("512x512 Y -i 50", 512, 512, "y", 50)):
    planes = synth.make_planes(W, H, "444" if sub == "y" else sub, 10, seed=1235, y_only=(sub == "y"))
    with j.Solver(planes, 0.3, [0.001] * len(planes), its) as s:
        def run():
            s.reset(); s.run(its); s.sync()
        for _ in range(5): run()
        t0 = time.perf_counter()
        for _ in range(40): run()
        out[name] = round((time.perf_counter() - t0) / 40 * 1e3, 4)
print(json.dumps(out))
PY
done 2>&1 | grep '^{'
