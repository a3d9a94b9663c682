#!/bin/bash
set -u
O=gpurun_out/r04c
mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_capi.py tests/test_cli.py -m gpu -x -q -k "drop_in or capi or reference_program" ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for t in 0 4; do J2P_COMPUTE_TIMING=1 J2P_XFER_THREADS=$t timeout 300 python - <<PY
import json, os, sys
sys.path.insert(0, ".")
import jpeg2png_amd as j
from jpeg2png_amd import synth
p = synth.make_planes(4096, 4096, "444", 10, seed=1237, y_only=True)
p[0].fdata = j.decode_plane(p[0])
_, secs = j.compute_c(p, 0.3, [0.001], 500, repeat=6)
print(json.dumps({"J2P_XFER_THREADS": os.environ["J2P_XFER_THREADS"], "ms_per_call": [round(s*1e3,2) for s in secs]}))
PY
done 2>&1 | grep -E '^\{|timing' | tee $O/xfer_timing2.txt
