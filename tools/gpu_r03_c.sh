#!/bin/bash
# round 3, call C: the cheaper source-term arithmetic (parity first), taller strips, a deeper row ring for mid-size
# planes, wave timelines without the shared counter
set -u
O=gpurun_out/r03c
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_baseline_configs_gpu.py tests/test_tiled_c_gpu.py -m gpu -x -q -k "not config2 and not full_size" ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.log
one() {  # label, env..., -- bench args
  python - "$@" <<'PY'
import json, os, subprocess, sys
label = sys.argv[1]
i = sys.argv.index("--")
env = dict(os.environ)
for kv in sys.argv[2:i]:
    k, v = kv.split("=", 1); env[k] = v
r = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-other-configs", *sys.argv[i + 1:]], capture_output=True, text=True, env=env)
line = [l for l in r.stdout.splitlines() if l.startswith("{")]
if not line:
    print(json.dumps({"label": label, "error": (r.stderr or r.stdout)[-300:]})); sys.exit(0)
d = json.loads(line[-1]); ro = d["roofline"]
print(json.dumps({"label": label, "workload": d["config"]["workload"][:40], "Mpx_it_per_s": d["value"], "us_per_iteration": round(ro["iteration_ms"] * 1e3, 2), "frac": ro["frac"],
                  "k_gradient_us": round(ro["per_kernel"]["k_gradient"]["avg_launch_ms"] * 1e3, 1), "k_project_us": round(ro["per_kernel"]["k_project"]["avg_launch_ms"] * 1e3, 1)}))
PY
}
{
one base4096 -- --steps 4 --warmup 2
one base4096_again -- --steps 4 --warmup 2
for r in 32 64; do one rpw${r}_4096 J2P_RPW=$r -- --steps 3 --warmup 1; done
one base_16384x2048 -- --size 16384 --height 2048 --iterations 100 --steps 3 --warmup 1
for r in 32 64; do one rpw${r}_16384x2048 J2P_RPW=$r -- --size 16384 --height 2048 --iterations 100 --steps 3 --warmup 1; done
for sz in "1920 1080" "2048 2048" "4096 2048" "4096 4096"; do
  set -- $sz
  one base_$1x$2 -- --size $1 --height $2 --iterations 100 --steps 3 --warmup 1
  one ring5_$1x$2 J2P_LIBRARY=variants/libj2p_ring5.so -- --size $1 --height $2 --iterations 100 --steps 3 --warmup 1
done
} | tee $O/ab.jsonl
for c in "512 512 420 rgb" "1920 1080 444 y" "2048 2048 444 y" "4096 4096 444 y"; do
  ( J2P_LIBRARY=variants/libj2p_trace.so timeout 120 python tools/wave_trace.py $c ) 2>&1 | grep '^{' | tee -a $O/wave_trace.jsonl
done
