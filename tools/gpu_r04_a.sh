#!/bin/bash
# round 4, first GPU call: the new band exchanges, batch / CLI changes, staged host copies
set -u
O=gpurun_out/r04a
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_tiled_c_gpu.py tests/test_batch_gpu.py tests/test_cli.py tests/test_parity_gpu.py tests/test_capi.py -m gpu -x -q --durations=5 ) > $O/pytest_a.log 2>&1; echo "pytest a rc=$?"; tail -15 $O/pytest_a.log
( timeout 600 python -m pytest tests/test_baseline_configs_gpu.py -m gpu -x -q -k "config0 or config1 or config4 or 16384_wide" --durations=5 ) > $O/pytest_b.log 2>&1; echo "pytest b rc=$?"; tail -15 $O/pytest_b.log
for combo in "direct root" "copy root" "copy all"; do set -- $combo; J2P_TILED_EXCHANGE=$1 J2P_TILED_NORM=$2 timeout 300 python tools/band_alone.py; done 2>&1 | grep '^{' | tee $O/band_alone.jsonl
timeout 300 python tools/band_nip.py 2>&1 | grep '^{' | tee $O/band_nip.json
( timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench.log 2>&1; echo "bench rc=$?"; grep '^{' $O/bench.log | tail -1 > $O/bench.json; python - <<PY
import json
d=json.load(open("$O/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:v["avg_launch_ms"] for k,v in d["roofline"]["per_kernel"].items()})
print(json.dumps(d.get("host_to_host")))
for o in d.get("other_configs",[]): print(json.dumps(o)[:300])
PY
for t in 0 1 2 4 8; do J2P_XFER_THREADS=$t timeout 300 python - <<PY
import json, os, sys
sys.path.insert(0, ".")
import jpeg2png_amd as j
from jpeg2png_amd import synth
p = synth.make_planes(4096, 4096, "444", 10, seed=1237, y_only=True)
p[0].fdata = j.decode_plane(p[0])
_, secs = j.compute_c(p, 0.3, [0.001], 500, repeat=4)
print(json.dumps({"J2P_XFER_THREADS": os.environ["J2P_XFER_THREADS"], "ms_per_call": [round(s*1e3,2) for s in secs]}))
PY
done 2>&1 | grep '^{' | tee $O/xfer_threads.jsonl
( timeout 900 python bench.py --force-tiled --bands 8 --steps 2 --warmup 1 --no-cpu-baseline ) 2>&1 | grep '^{' | tail -1 > $O/bench_tiled_8bands_1gpu.json; cut -c1-1500 $O/bench_tiled_8bands_1gpu.json
