import json, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for n in [int(x) for x in sys.argv[1:]]:
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--size", str(n), "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    t = d["roofline"]["avg_launch_ms"]
    print(n, d["value"], t, "ps/px grad %.3f proj %.3f" % (t["k_gradient"] * 1e9 / (n * n), t["k_project"] * 1e9 / (n * n)), flush=True)
