#!/bin/bash
set -u
O=gpurun_out/r03f; mkdir -p $O; export TMPDIR=/tmp
for combo in "1 1 root" "1 0 root" "0 0 root" "0 0 all" "1 0 all"; do
  set -- $combo
  J2P_TILED_SPLIT=$1 J2P_TILED_SIDE=$2 J2P_TILED_NORM=$3 timeout 300 python tools/band_alone.py | tee -a $O/band_alone.jsonl
done
for combo in "1 0 root" "0 0 root" "0 0 all"; do
  set -- $combo
  ( J2P_TILED_SPLIT=$1 J2P_TILED_SIDE=$2 J2P_TILED_NORM=$3 timeout 600 python bench.py --force-tiled --bands 8 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --tiled-impl c ) 2>&1 | grep '^{' | tail -1 > $O/tmp.json
  python - <<PY
import json
d=json.load(open("$O/tmp.json"))
print(json.dumps({"bands_on_one_gpu": 8, "split": $1, "side": $2, "norm": "$3", "Mpx_it_per_s": d["value"], "ms_per_step": d["ms_per_step"], "host_cpu_s": d["config"].get("band_threads_host_cpu_s")}))
PY
done | tee $O/bands8.jsonl
