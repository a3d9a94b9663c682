#!/bin/bash
# round 3, call B: instruction issue costs, wave timelines of small / mid-size planes, rows-per-strip sweep,
# what the split phases cost a band that has a GPU to itself
set -u
O=gpurun_out/r03b
mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/ubench/valu_rates > $O/valu_rates.json; cat $O/valu_rates.json
for c in "512 512 420 rgb" "1920 1080 444 y" "1920 1080 420 rgb" "2048 2048 444 y" "4096 4096 444 y"; do
  ( J2P_LIBRARY=variants/libj2p_trace.so timeout 120 python tools/wave_trace.py $c ) 2>&1 | grep '^{' | tee -a $O/wave_trace.jsonl
done
for sz in "1920 1080" "2048 2048" "4096 2048" "4096 4096"; do
  set -- $sz
  for r in 4 8 16; do
    ( J2P_RPW=$r timeout 120 python bench.py --size $1 --height $2 --iterations 100 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs ) 2>&1 | grep '^{' | tail -1 > $O/tmp.json
    python - <<PY
import json
d=json.load(open("$O/tmp.json")); r=d["roofline"]
print(json.dumps({"plane":"$1x$2","rpw":$r,"Mpx_it_per_s":d["value"],"us_per_iteration":round(r["iteration_ms"]*1e3,2),"frac":r["frac"],"k_gradient_us":round(r["per_kernel"]["k_gradient"]["avg_launch_ms"]*1e3,1),"k_project_us":round(r["per_kernel"]["k_project"]["avg_launch_ms"]*1e3,1)}))
PY
  done
done | tee $O/rpw_sweep.jsonl
# a 2048-row band next to a 48-row band: band 0 has the GPU practically to itself, so its iteration time is what
# the split phases (interior / edges, boundary / interior, halo copy, norm) cost on a dedicated GPU
python - <<'PY' | tee $O/band_alone.json
import json, time
import jpeg2png_amd as j
from jpeg2png_amd import synth
W, its = 16384, 100
res = {}
p = synth.make_y_plane_banded(W, 2096, 10, 1238, band_rows=1048 - 1048 % 8, workers=2) if False else synth.make_planes(W, 2096, "444", 10, seed=1238, y_only=True)[0]
def timed(fn, reps=3):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps
with j.TiledSolver([p], 0.3, [0.001], its, devices=[0, 0], cuts=[0, 2048, 2096]) as t:
    def run():
        t.reset(); t.run(its); t.sync()
    res["band_2048_plus_band_48_us_per_iteration"] = round(timed(run) / its * 1e6, 2)
with j.Solver([p], 0.3, [0.001], its) as s:
    def run():
        s.reset(); s.run(its); s.sync()
    res["whole_2096_rows_us_per_iteration"] = round(timed(run) / its * 1e6, 2)
print(json.dumps(res))
PY
