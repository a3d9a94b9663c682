#!/usr/bin/env python3
"""one case of the band sweep under schedule switches, with a per-iteration look at every band's planes:
   python tools/dbg_bands.py SEED INDEX [mixed=0|1] [log=0|1]"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import jpeg2png_amd as j
from sweep_cases import cases

seed, index = int(sys.argv[1]), int(sys.argv[2])
opts = dict(kv.split("=") for kv in sys.argv[3:])
mixed, log = int(opts.get("mixed", 1)), int(opts.get("log", 1))
edges = [int(v) for v in opts["edges"].split(",")]
cs = list(cases(seed, index + 1))[index]
print(cs.describe(), "edges", edges, "mixed", mixed, "log", log)
planes = cs.planes()
for p in planes:
    p.fdata = j.decode_plane(p)
nch = len(planes)
its = min(cs.iterations, 4)
hip = j.hip_runtime()
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
whole = j.Solver(planes, cs.weight, cs.pweights, its)
whole.debug_option(j.J2P_OPT_MIXED_PROJECT, mixed)
bands = [j.Solver(planes, cs.weight, cs.pweights, its, band=(edges[i], edges[i + 1])) for i in range(len(edges) - 1)]
for s in bands:
    s.debug_option(j.J2P_OPT_MIXED_PROJECT, mixed)
    s.set_logging(bool(log))
for it in range(its):
    whole.run(1)
    whole.sync()
    for s in bands:
        s.phase_gradient()
    infos = [s.exchange_info() for s in bands]
    for s in bands:
        s.sync()
    for dst in infos:
        for src in infos:
            hip.hipMemcpy(dst.partials_all + 8 * nch * src.first_tile_row, src.partials_local, 8 * nch * src.local_tile_rows, 3)
    for s in bands:
        s.phase_project()
    for s in bands:
        s.sync()
    infos = [s.exchange_info() for s in bands]
    nbytes = infos[0].halo_floats * 4
    for i in range(len(bands) - 1):
        for c in range(nch):
            hip.hipMemcpy(infos[i + 1].recv_top[c], infos[i].send_bottom[c], nbytes, 3)
            hip.hipMemcpy(infos[i].recv_bottom[c], infos[i + 1].send_top[c], nbytes, 3)
    for bi, info in enumerate(infos):
        part = np.zeros(5)
        if log:
            hip.hipMemcpy(part.ctypes.data, info.log_local, 40, 2)
        print(f"it {it} band {bi} log sums {part}")
    for c in range(nch):
        w = whole.download(c)
        for bi, s in enumerate(bands):
            g = s.download(c)
            ref = w[edges[bi]:edges[bi + 1]]
            neq = np.argwhere(g.view(np.uint32) != ref.view(np.uint32))
            print(f"it {it} ch {c} band {bi}: nan {int(np.isnan(g).sum())} whole-nan {int(np.isnan(ref).sum())} differing {len(neq)}"
                  + (f" first at row {neq[0][0]} col {neq[0][1]} (cols {neq[:, 1].min()}..{neq[:, 1].max()}, rows {neq[:, 0].min()}..{neq[:, 0].max()})" if len(neq) else ""))
