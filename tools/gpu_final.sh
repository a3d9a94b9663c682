#!/bin/bash
# the round's evidence in one call: whole GPU suite, profiles (kernel stats, counters, bench line), the multi-rank bench
# control flow on one GPU, the row tiling and the batch engine on one GPU, sweeps on the final kernels
set -u
TAG=${1:-r05}
O=gpurun_out/${TAG}_final
mkdir -p $O
export TMPDIR=/tmp
if [ "${SKIP_SUITE:-0}" != 1 ]; then ( timeout 1500 python -m pytest tests -m gpu -q --durations=8 --timeout 900 ) > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -26 $O/pytest_gpu.log; fi
( timeout 300 python __graft_entry__.py --smoke ) 2>&1 | tail -1
bash tools/collect_profiles.sh $TAG 2>&1 | tail -6
# the line as the driver runs it (BENCH_rNN.json: `python3 bench.py --gpus 1 --steps 20 --warmup 5`)
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) 2>/dev/null | grep '^{' | tail -1 > gpurun_out/${TAG}_bench_driver_shape.json; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_driver_shape.json')); r=d['roofline']; h=d.get('host_to_host',{})
print('driver shape:', d['value'], d['ms_per_step'], r['frac'], d['parity']['bit_identical'], h.get('ms_per_call'), h.get('ms_per_call_in_call_order'), [o.get('Mpx_it_per_s') for o in d.get('other_configs',[])])"
# two ranks on this box's one GPU (gloo): the driver's --gpus N launch shape, rank 0 driving two bands, both C schedules
( J2P_BENCH_ONE_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --size 4096 ) > $O/bench_2ranks.log 2>&1; echo "2-rank bench rc=$?"; grep '^{' $O/bench_2ranks.log | tail -1 > gpurun_out/${TAG}_bench_2ranks_1gpu_gloo.json; cut -c1-400 gpurun_out/${TAG}_bench_2ranks_1gpu_gloo.json
# 8 bands of 16384x2048 on one GPU (C engine, both schedules; strong-scaling denominator; 256-image batch)
( timeout 600 python bench.py --force-tiled --bands 8 --steps 2 --warmup 1 --no-cpu-baseline ) 2>&1 | grep '^{' | tail -1 > gpurun_out/${TAG}_bench_tiled_8bands_1gpu.json; cut -c1-300 gpurun_out/${TAG}_bench_tiled_8bands_1gpu.json
# the RCCL exchange of the C engine with one band as its own neighbour (ncclCommInitAll, ncclAllGather, grouped
# ncclSend / ncclRecv on real hardware), against the same rows solved whole
( J2P_LIBRARY=jpeg2png_amd/libjpeg2png_amd_exp.so J2P_TILED_EXCHANGE=rccl J2P_TILED_SELF_NEIGHBOURS=1 timeout 300 python - <<PY
import json, sys, time
sys.path.insert(0, ".")
import jpeg2png_amd as j
from jpeg2png_amd import synth
p = synth.make_planes(16384, 2048, "444", 10, seed=1238, y_only=True)[0]
its = 100
def timed(fn, reps=3):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps
with j.TiledSolver([p], 0.3, [0.001], its, devices=[0]) as t:
    def run():
        t.reset(); t.run(its); t.sync()
    a = timed(run); ex = t.exchange()
with j.Solver([p], 0.3, [0.001], its) as s:
    def run():
        s.reset(); s.run(its); s.sync()
    b = timed(run)
print(json.dumps({"c_engine_exchange": ex, "one_band_16384x2048_as_its_own_neighbour_us_per_iteration": round(a / its * 1e6, 2), "whole_us_per_iteration": round(b / its * 1e6, 2)}))
PY
) 2>&1 | grep '^{' | tee gpurun_out/${TAG}_rccl_self_neighbours.jsonl
( timeout 300 python bench.py --config batch --steps 2 --warmup 1 --batch 32 ) 2>&1 | grep '^{' | tee gpurun_out/${TAG}_bench_batch.json | cut -c1-300
for combo in "direct root" "copy root" "copy all"; do set -- $combo; J2P_TILED_EXCHANGE=$1 J2P_TILED_NORM=$2 timeout 300 python tools/band_alone.py; done 2>&1 | grep '^{' | tee gpurun_out/${TAG}_band_alone.jsonl
timeout 300 python tools/nt_scope.py | tee gpurun_out/${TAG}_nt_scope.jsonl
# size sweep on the final kernels
for sz in "1920 1080" "2048 2048" "4096 2048" "4096 4096" "4096 5120" "8192 4096" "16384 2048" "8192 8192" "16384 4096" "16384 8192"; do
  set -- $sz
  ( timeout 200 python bench.py --size $1 --height $2 --iterations 100 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs ) 2>&1 | grep '^{' | tail -1 > $O/tmp.json
  python - <<PY
import json
d=json.load(open("$O/tmp.json")); r=d["roofline"]
k=r["per_kernel"]
print(json.dumps({"plane":"$1x$2 Y-only Q10 -i 100","Mpx_it_per_s":d["value"] or d.get("unverified_value"),"us_per_iteration":round(r["iteration_ms"]*1e3,2),"iteration_frac":r["frac"],"k_gradient_us":round(k["k_gradient"]["avg_launch_ms"]*1e3,1) if "k_gradient" in k else None,"k_project_us":round(k["k_project"]["avg_launch_ms"]*1e3,1) if "k_project" in k else None,"parity":(d.get("parity") or {}).get("bit_identical")}))
PY
done | tee gpurun_out/${TAG}_size_sweep.jsonl
