#!/bin/bash
# same-box A/B: round 3's library (ab/r03tree, built from commit a770bed) against the current one: headline workload and the
# other configurations of the bench line.  ab/r03tree is made here, before the call (ab/ travels with the lease):
#   mkdir -p /tmp/r03tree ab && git archive a770bed jpeg2png_amd include bench.py oracle/bindings.py oracle/__init__.py | tar -x -C /tmp/r03tree
#   (cd /tmp/r03tree && python -c "import jpeg2png_amd; jpeg2png_amd.build()") && cp -r /tmp/r03tree ab/r03tree
set -u
O=gpurun_out/r04f
mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
for rep in 1 2; do
  for which in r03 cur; do
    if [ $which = r03 ]; then D=ab/r03tree; EXTRA=""; else D=.; EXTRA="--no-host-to-host"; fi
    ( cd $D && timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $EXTRA ) 2>&1 | grep '^{' | tail -1 > $R/$O/tmp.json
    python - <<PY
import json
d=json.load(open("$R/$O/tmp.json")); r=d["roofline"]
oc=d["other_configs"]
print(json.dumps({"library":"$which","rep":$rep,"Mpx_it_per_s":d["value"],"us_per_iteration":round(r["iteration_ms"]*1e3,2),"config0_ms":oc[0]["ms_per_solve"],"config1_Mpx":oc[1]["Mpx_it_per_s"],"config4_slice_images_per_s":oc[2]["images_per_s"],"band_16384x2048_us":oc[3]["us_per_iteration"]}))
PY
  done
done | tee $O/ab_r03_vs_r04_all.jsonl
