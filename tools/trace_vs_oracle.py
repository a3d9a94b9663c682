"""diagnostic: run sweep case N phase by phase on the GPU and compare the gradient and the iterate of every
iteration with the CPU oracle's trace (bitwise, so also the sign of zero).
usage: python tools/trace_vs_oracle.py N ITS [seed]"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
import jpeg2png_amd as j
from jpeg2png_amd import synth
from oracle import bindings as oracle

target, its_override = int(sys.argv[1]), int(sys.argv[2])
sys.path.insert(0, "tests")
from sweep_cases import case
cs = case(int(sys.argv[3]) if len(sys.argv) > 3 else 2, target)
W, H, sub, q, y_only, its, weight, pws = cs.W, cs.H, cs.sub, cs.quality, cs.y_only, cs.iterations, cs.weight, cs.pweights
its = its_override
planes = cs.planes()
for p in planes:
    p.fdata = oracle.decode_plane(p)
nch = len(planes)
CW, CH = oracle.canvas_size(planes)
trace = np.zeros((its, nch, 2, CH, CW), dtype=np.float32)
lib = oracle.oracle_lib()
lib.oracle_set_trace.argtypes = [ctypes.c_void_p]
lib.oracle_set_trace(trace.ctypes.data)
oracle.oracle_compute(planes, weight, pws, its)
lib.oracle_set_trace(None)
np.set_printoptions(linewidth=250, precision=4)
s = j.Solver(planes, weight, pws, its)
for it in range(its):
    s.phase_gradient()
    for what, idx in (("gradient", 0), ("iterate", 1)):
        if what == "iterate":
            s.phase_project()
        for c in range(nch):
            got = s.download_gradient(c) if what == "gradient" else s.download(c)
            want = trace[it, c, idx]
            dm = got.view(np.uint32) != want.view(np.uint32)
            if dm.any():
                ys, xs = np.nonzero(dm)
                y0, x0 = int(ys[0]), int(xs[0])
                print(f"iteration {it} {what} channel {c}: {len(ys)} differ; first ({x0},{y0}) gpu {got[y0, x0]!r} {hex(int(got.view(np.uint32)[y0, x0]))} oracle {want[y0, x0]!r} {hex(int(want.view(np.uint32)[y0, x0]))}")
                by, bx = y0 // 8 * 8, x0 // 8 * 8
                if what == "iterate":
                    print(" oracle gradient block:\n", trace[it, c, 0][by:by + 8, bx:bx + 8])
                    if it:
                        print(" x_k block:\n", trace[it - 1, c, 1][by:by + 8, bx:bx + 8])
                    if it > 1:
                        print(" x_{k-1} block:\n", trace[it - 2, c, 1][by:by + 8, bx:bx + 8])
                    print(" oracle new block:\n", want[by:by + 8, bx:bx + 8])
                    print(" gpu new block:\n", got[by:by + 8, bx:bx + 8])
                    # redo the block's step + projection with numpy float32 and the standalone DCT entry points
                    gfull = trace[it, c, 0]
                    norm = np.sqrt(np.float32(np.sum(gfull.astype(np.float64) ** 2)))
                    step = np.float32(np.sqrt(np.float32(CH) * np.float32(CW)) / np.float32(2)) / np.sqrt(np.float32(1 + its))
                    xk = trace[it - 1, c, 1][by:by + 8, bx:bx + 8]
                    gb = gfull[by:by + 8, bx:bx + 8]
                    stepped = (xk - step * (gb / norm)).astype(np.float32)      # factor*(x_k - x_{k-1}) = 0 here
                    print(" norm", norm, "step", step, "stepped block:\n", stepped)
                    blk = stepped.reshape(1, 64)
                    for name, f in (("gpu", lambda b, inv: j.dct8x8_blocks(b, inverse=inv)), ("cpu", lambda b, inv: oracle.dct_blocks(b, inverse=inv))):
                        co = f(blk, False)
                        back = f(co, True)
                        print(f" standalone {name}: coefficients nonzero {int((co != 0).sum())}, back signs-of-zero:",
                              [(int(i % 8), int(i // 8)) for i in np.nonzero(back.view(np.uint32)[0] == 0x80000000)[0]])
                        print(back.reshape(8, 8))
                else:
                    print(" oracle:\n", want[max(0, y0 - 2):y0 + 3, max(0, x0 - 2):x0 + 3], "\n gpu:\n", got[max(0, y0 - 2):y0 + 3, max(0, x0 - 2):x0 + 3])
                sys.exit(1)
print("identical through", its, "iterations")
