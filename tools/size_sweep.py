import json, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for arg in sys.argv[1:]:
    w, h = (int(x) for x in arg.split("x")) if "x" in arg else (int(arg), int(arg))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--size", str(w), "--height", str(h), "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    t = d["roofline"]["avg_launch_ms"]
    print(f"{w}x{h}", d["value"], t, "ps/px grad %.3f proj %.3f" % (t["k_gradient"] * 1e9 / (w * h), t["k_project"] * 1e9 / (w * h)),
          flush=True)
