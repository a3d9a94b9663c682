"""randomised sweep of the row-band solver: 2-4 bands on ONE GPU, halos and norm partials exchanged by device
copies, must reproduce the whole-canvas solver bit for bit whatever the cut positions (the reduction order is
GPU-count invariant).  usage: python tools/sweep_bands.py [ncases] [seed]"""
import copy
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import jpeg2png_amd as j
from jpeg2png_amd import tiled
from sweep_cases import cases

split = "--split" in sys.argv           # use the two-part projection phase, halo rows copied between the parts
if split:
    sys.argv.remove("--split")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
hip = j.hip_runtime()
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
D2D = 3


def d2d_done():
    """hipMemcpy device-to-device goes to the NULL stream and may return before the copy has happened, and the
    solvers' streams are non-blocking ones: without this the next phase can read a norm partial or a halo row
    that has not arrived yet (seen once in ~10 runs of the 25-case test as NaN log rows)"""
    hip.hipDeviceSynchronize()


rng = np.random.default_rng([seed, 77])
bad = ran = 0
for cs in cases(seed, n):
    planes = cs.planes()
    for p in planes:
        p.fdata = j.decode_plane(p)
    nch = len(planes)
    align = tiled.band_alignment(planes)
    H = max(p.h * p.h_samp for p in planes)
    units = (H + align - 1) // align
    if units < 2:
        continue
    nb = int(rng.integers(2, min(4, units) + 1))
    cuts = sorted(rng.choice(np.arange(1, units), nb - 1, replace=False).tolist())
    edges = [0] + [c * align for c in cuts] + [H]
    its = min(cs.iterations, 12)
    ran += 1
    ref = copy.deepcopy(planes)
    ref_rows = j.compute(ref, cs.weight, cs.pweights, its, log=True)
    bands = [j.Solver(planes, cs.weight, cs.pweights, its, band=(edges[i], edges[i + 1])) for i in range(nb)]
    sums = np.zeros((its, 5))
    try:
        for s in bands:
            s.set_logging(True)
        for it in range(its):
            for s in bands:
                s.phase_gradient()
            infos = [s.exchange_info() for s in bands]
            for s in bands:
                s.sync()
            for dst in infos:                      # all-gather of the [tile_row][channel] partials
                for src in infos:
                    hip.hipMemcpy(dst.partials_all + 8 * nch * src.first_tile_row, src.partials_local,
                                  8 * nch * src.local_tile_rows, D2D)
            d2d_done()
            if split:                              # boundary block rows first, halo copy, then the rest
                for s in bands:
                    s.phase_project_part(1)
                for s in bands:
                    s.sync()
                infos = [s.exchange_info() for s in bands]      # now points into the iterate being written
                nbytes = infos[0].halo_floats * 4
                for i in range(nb - 1):
                    for c in range(nch):
                        hip.hipMemcpy(infos[i + 1].recv_top[c], infos[i].send_bottom[c], nbytes, D2D)
                        hip.hipMemcpy(infos[i].recv_bottom[c], infos[i + 1].send_top[c], nbytes, D2D)
                d2d_done()
                for s in bands:
                    s.phase_project_part(2)
            else:
                for s in bands:
                    s.phase_project()
            for s in bands:
                s.sync()
            infos = [s.exchange_info() for s in bands]
            for info in infos:                     # the bands' tv / tv2 / prob sums of this iteration
                part = np.zeros(5)
                hip.hipMemcpy(part.ctypes.data, info.log_local, 40, 2)
                if not np.all(np.isfinite(part)):
                    print(f"      non-finite band log sums at iteration {it}, band {infos.index(info)}: {part}")
                sums[it] += part
            nbytes = infos[0].halo_floats * 4
            for i in range(nb - 1):
                for c in range(nch):
                    if split:
                        break
                    hip.hipMemcpy(infos[i + 1].recv_top[c], infos[i].send_bottom[c], nbytes, D2D)
                    hip.hipMemcpy(infos[i].recv_bottom[c], infos[i + 1].send_top[c], nbytes, D2D)
            d2d_done()
        same = True
        for c in range(nch):
            got = np.concatenate([s.download(c) for s in bands], axis=0)
            same &= np.array_equal(got.view(np.uint32), ref[c].fdata.view(np.uint32))
    finally:
        for s in bands:
            s.close()
    rows = j.log_rows_from_sums(nch, cs.weight, cs.pweights, sums)
    if its and not np.allclose(rows, ref_rows, rtol=1e-9, atol=1e-12):
        same = False
        print("      log rows differ:", rows[:2], ref_rows[:2])
    bad += not same
    print(("ok   " if same else "DIFF ") + cs.describe() + f"  bands {edges} its {its}", flush=True)
print(f"{ran - bad}/{ran} band splits bit-identical to the whole canvas")
sys.exit(1 if bad else 0)
