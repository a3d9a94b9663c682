"""randomised end-to-end sweep: cli/jpeg2png_gpu vs the UNMODIFIED reference program on random JPEGs (size,
quality, subsampling, progressive, restart markers) and random flags; the PNGs must be byte-identical.
usage: python tools/sweep_cli.py [ncases] [seed]      (needs oracle/_ref/jpeg2png_ref and PIL)"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from PIL import Image
from jpeg2png_amd import synth
from jpeg2png_amd.buildlib import build_cli

REF = os.path.join(ROOT, "oracle", "_ref", "jpeg2png_ref")
exe = build_cli()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
with tempfile.TemporaryDirectory() as tmp:
    for i in range(n):
        w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500))
        if rng.random() < 0.25:
            w, h = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        q = int(rng.choice([3, 10, 25, 50, 75, 90, 100]))
        sub = int(rng.choice([0, 1, 2]))
        kw = {}
        if rng.random() < 0.3:
            kw["progressive"] = True
        if rng.random() < 0.3:
            kw["optimize"] = True
        jpg = os.path.join(tmp, f"c{i}.jpg")
        rgb = synth.synth_rgb(w, h, int(rng.integers(1 << 30))).astype(np.uint8)
        if rng.random() < 0.2:
            rgb[: h // 2, : w // 2] = 128                       # flat grey area
        Image.fromarray(rgb, "RGB").save(jpg, "JPEG", quality=q, subsampling=sub, **kw)
        flags = ["-i", str(int(rng.integers(0, 25)))]
        if rng.random() < 0.3:
            flags = ["-s", "-i", ",".join(str(int(x)) for x in rng.integers(0, 15, 3)),
                     "-w", ",".join(str(float(x)) for x in rng.choice([0.0, 0.1, 0.3, 1.0], 3))]
        elif rng.random() < 0.5:
            flags += ["-w", str(float(rng.choice([0.0, 0.1, 0.3, 1.0])))]
        if rng.random() < 0.4:
            flags += ["-p", str(float(rng.choice([0.0, 0.001, 0.01])))]
        if rng.random() < 0.3:
            flags += ["-1"]
        ref_png, gpu_png = os.path.join(tmp, f"r{i}.png"), os.path.join(tmp, f"g{i}.png")
        r = subprocess.run([REF, jpg, "-o", ref_png, "-q", "-t", "1", *flags], capture_output=True, text=True)
        g = subprocess.run([exe, jpg, "-o", gpu_png, "-q", *flags], capture_output=True, text=True)
        ok = r.returncode == g.returncode and (r.returncode != 0 or open(ref_png, "rb").read() == open(gpu_png, "rb").read())
        if r.returncode != 0:
            ok = ok and r.stderr.strip() == g.stderr.strip()
        bad += not ok
        print(("ok   " if ok else "DIFF ") + f"{i:3d} {w}x{h} q{q} sub{sub} {kw} {' '.join(flags)}"
              + ("" if ok else f"  rc {r.returncode}/{g.returncode} ref: {r.stderr.strip()[:100]} gpu: {g.stderr.strip()[:100]}"), flush=True)
print(f"{n - bad}/{n} PNGs byte-identical to the reference program's")
sys.exit(1 if bad else 0)
