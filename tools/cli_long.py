"""CLI with the default iteration count (50, no -i) and with more iterations than one host chunk (-i 70, 100)"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from PIL import Image
from jpeg2png_amd import synth
from jpeg2png_amd.buildlib import build_cli
REF = os.path.join(ROOT, "oracle", "_ref", "jpeg2png_ref")
exe = build_cli()
bad = 0
with tempfile.TemporaryDirectory() as tmp:
    for i, (w, h, q, sub, flags) in enumerate([(200, 150, 30, 2, []), (123, 77, 10, 0, ["-i", "70"]), (96, 200, 50, 1, ["-i", "100", "-c", "CSV"]),
                                              (64, 64, 20, 2, ["-s"]), (150, 90, 15, 2, ["-s", "-i", "70,40,33", "-c", "CSV"])]):
        jpg = os.path.join(tmp, f"c{i}.jpg")
        Image.fromarray(synth.synth_rgb(w, h, 100 + i).astype(np.uint8), "RGB").save(jpg, "JPEG", quality=q, subsampling=sub)
        outs = []
        for tag, prog, extra in (("ref", REF, ["-t", "1"]), ("gpu", exe, [])):
            png = os.path.join(tmp, f"{tag}{i}.png")
            fl = [f if f != "CSV" else os.path.join(tmp, f"{tag}{i}.csv") for f in flags]
            r = subprocess.run([prog, jpg, "-o", png, "-q", *extra, *fl], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            outs.append(open(png, "rb").read())
        ok = outs[0] == outs[1]
        if "CSV" in flags:
            a = np.loadtxt(os.path.join(tmp, f"ref{i}.csv"), delimiter=",", skiprows=1, usecols=(1, 2, 3, 4, 5, 6), ndmin=2)
            b = np.loadtxt(os.path.join(tmp, f"gpu{i}.csv"), delimiter=",", skiprows=1, usecols=(1, 2, 3, 4, 5, 6), ndmin=2)
            a = a[np.lexsort((a[:, 1], a[:, 0]))]; b = b[np.lexsort((b[:, 1], b[:, 0]))]
            ok = ok and a.shape == b.shape and np.allclose(a, b, rtol=0, atol=2e-6 * max(1.0, np.abs(a).max()))
        bad += not ok
        print(("ok   " if ok else "DIFF ") + f"{w}x{h} q{q} sub{sub} {flags}", flush=True)
sys.exit(1 if bad else 0)
