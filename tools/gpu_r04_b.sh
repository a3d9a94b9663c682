#!/bin/bash
# round 4, second GPU call: host-to-host with the persistent copy team, the per-workgroup norm tree on whole canvases,
# the multi-process test, the checked build
set -u
O=gpurun_out/r04b
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_tiled_multiprocess_gpu.py tests/test_debug_build_gpu.py tests/test_tiled_c_gpu.py -m gpu -x -q --durations=5 ) > $O/pytest_a.log 2>&1; echo "pytest a rc=$?"; tail -12 $O/pytest_a.log
( timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "schedule_switch or drop_in" ) > $O/pytest_b.log 2>&1; echo "pytest b rc=$?"; tail -5 $O/pytest_b.log
for t in 0 2 4 8; do J2P_XFER_THREADS=$t timeout 300 python - <<PY
import json, os, sys
sys.path.insert(0, ".")
import jpeg2png_amd as j
from jpeg2png_amd import synth
p = synth.make_planes(4096, 4096, "444", 10, seed=1237, y_only=True)
p[0].fdata = j.decode_plane(p[0])
_, secs = j.compute_c(p, 0.3, [0.001], 500, repeat=5)
print(json.dumps({"J2P_XFER_THREADS": os.environ["J2P_XFER_THREADS"], "ms_per_call": [round(s*1e3,2) for s in secs]}))
PY
done 2>&1 | grep '^{' | tee $O/xfer_threads.jsonl
# where ||g|| is finished on WHOLE canvases: k_norm_whole launch (default above 2.5 Mpixel) / per-wavefront tree / per-workgroup tree
for sz in "1920 1080" "2048 2048" "4096 2048" "4096 4096" "16384 2048" "8192 8192"; do
  set -- $sz
  for mode in "-1 -1" "1 1" "1 2"; do
    set -- $sz $mode
    ( timeout 200 python bench.py --size $1 --height $2 --iterations 100 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-to-host --norm-fold $3 --norm-in-project $4 ) 2>&1 | grep '^{' | tail -1 > $O/tmp.json
    python - <<PY
import json
d=json.load(open("$O/tmp.json")); r=d["roofline"]
print(json.dumps({"plane":"$1x$2 Y-only Q10 -i 100","norm_fold":$3,"norm_in_project":$4,"Mpx_it_per_s":d["value"],"us_per_iteration":round(r["iteration_ms"]*1e3,2),"iteration_frac":r["frac"],"k_gradient_us":round(r["per_kernel"]["k_gradient"]["avg_launch_ms"]*1e3,1),"k_project_us":round(r["per_kernel"]["k_project"]["avg_launch_ms"]*1e3,1)}))
PY
  done
done | tee $O/nip_whole.jsonl
for mode in "-1 -1" "1 2" "-1 -1" "1 2"; do
  set -- $mode
  ( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-host-to-host --norm-fold $1 --norm-in-project $2 ) 2>&1 | grep '^{' | tail -1 > $O/tmp.json
  python - <<PY
import json
d=json.load(open("$O/tmp.json")); r=d["roofline"]
print(json.dumps({"headline 4096^2 -i 500, norm_fold":$1,"norm_in_project":$2,"Mpx_it_per_s":d["value"],"us_per_iteration":round(r["iteration_ms"]*1e3,2),"iteration_frac":r["frac"]}))
PY
done | tee $O/nip_headline.jsonl
