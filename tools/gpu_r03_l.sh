#!/bin/bash
set -u
O=gpurun_out/r03l; mkdir -p $O; export TMPDIR=/tmp
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
for i in 1 2 3; do
one base -- --steps 4 --warmup 2
one noslp J2P_LIBRARY=variants/libj2p_noslp.so -- --steps 4 --warmup 2
done
one fold -- --steps 4 --warmup 2 --norm-fold 1 --norm-in-project 0
one fold_nip -- --steps 4 --warmup 2 --norm-fold 1 --norm-in-project 1
one base_1080p -- --size 1920 --height 1080 --iterations 100 --steps 3 --warmup 1
one noslp_1080p J2P_LIBRARY=variants/libj2p_noslp.so -- --size 1920 --height 1080 --iterations 100 --steps 3 --warmup 1
one base_16384x2048 -- --size 16384 --height 2048 --iterations 100 --steps 3 --warmup 1
one noslp_16384x2048 J2P_LIBRARY=variants/libj2p_noslp.so -- --size 16384 --height 2048 --iterations 100 --steps 3 --warmup 1
} | tee $O/ab.jsonl
