set -u
O=gpurun_out/r03n; mkdir -p $O; export TMPDIR=/tmp
( true || timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_debug_build_gpu.py tests/test_tiled_c_gpu.py tests/test_baseline_configs_gpu.py -m gpu -x -q -k "not config2 and not full_size and not exhaustively and not short_division" ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
one() {
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
print(json.dumps({"label": label, "Mpx_it_per_s": d["value"], "us_per_iteration": round(ro["iteration_ms"] * 1e3, 2), "frac": ro["frac"],
                  "k_gradient_us": round(ro["per_kernel"]["k_gradient"]["avg_launch_ms"] * 1e3, 1), "k_project_us": round(ro["per_kernel"]["k_project"]["avg_launch_ms"] * 1e3, 1)}))
PY
}
{
for sz in "16384 2048" "4096 4096" "2048 2048"; do set -- $sz
for v in prev po1 po2 po3 cur; do
  if [ $v = cur ]; then one ${v}_$1x$2 -- --size $1 --height $2 --iterations 100 --steps 3 --warmup 1
  else one ${v}_$1x$2 J2P_LIBRARY=variants/libj2p_$v.so -- --size $1 --height $2 --iterations 100 --steps 3 --warmup 1; fi
done; done
} | tee $O/ab3.jsonl
