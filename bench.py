#!/usr/bin/env python3
"""bench.py — Mpixel-iterations/s of the jpeg2png deblocking solver on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`
prints ONE JSON line on rank 0.  A *step* is one complete solve of the workload —
reset to iteration 0 from inputs already resident in HBM, then all its iterations
(the step size radius/sqrt(1+iterations) ties the iterations of one solve together,
compute.c:443) — with no host copies inside the timed region.

Workloads (BASELINE.json configs):
  N = 1 : configs[2]  4096x4096 Y-only, Q=10, -i 500, weight 0.3, pweight 0.001
          (the configuration the metric "Mpixel-iterations/sec on 4K Y-plane" is quoted on).
          `other_configs` carries configs[0], configs[1], a configs[4] slice and the 16384x2048 band that is
          the N = 1 point of the N > 1 workload below.
  N > 1 : configs[3]  16384-wide Y-only plane, Q=10, -i 100, row-tiled: 2048 rows per GPU
          (N = 8 is exactly the 16384x16384 config); weak scaling (fixed rows per GPU).  Every engine is
          timed, each with W warm-up and K timed steps between barriers, and the plane each leaves is hashed:
            c      : (and c_waitroot, c_waitcoll: the same with the projection waiting for one event of a root band / of a
                     collecting stream instead of N - 1 events)
                     the C row tiling (j2p_tiled: one process drives all N GPUs with one host thread per band) in its
                     default exchange — "direct": row sums of g^2 pushed from k_gradient and edge rows from k_project
                     as posted peer writes over xGMI, ||g|| reduced inside k_project, two launches and two event waits
                     per band and iteration — rank 0 drives it; the other ranks of the launch wait in a gloo (CPU)
                     barrier, so that no RCCL barrier kernel spins on their GPUs while rank 0's band kernels run there;
            c_copy : the same engine in round 3's exchange (copy kernel pulls the edge rows, one band reduces ||g|| for
                     all): four launches and three hops per iteration — the cross-check of `c` on real hardware;
            c_rccl : the same engine over RCCL (ncclAllGather + grouped ncclSend / ncclRecv on the band streams, the
                     exchange north_star names), librccl dlopen()ed by the C library;
            rccl   : one process per GPU, the Python harness over torch.distributed / librccl (jpeg2png_amd/tiled.py).
          `value` is the fastest leg whose plane has the same hash as the SAME canvas solved whole on ONE GPU
          (`other_configs` carries that solve — the strong-scaling denominator — and every other leg, each with
          `bits_equal_to_the_whole_canvas_solve`), and configs[4] — 256 x 1080p 4:2:0 Q50 -i 100 through the C
          batch engine over all N GPUs.
  --config batch : configs[4] slice — B x 1080p 4:2:0 Q=50 -i 100 through the C batch API
          (host buffers in, RGB out: PCIe inclusive), images/s and Mpx-it/s.

value = canvas pixels x iterations x steps / wall time over all ranks (max over ranks).
The JSON also carries
  roofline     : WHOLE ITERATION, wall clock: 38 algorithmic bytes per pixel-iteration
                 (gradient 16 B/px + step/projection 22 B/px) x px-it/s against the
                 8 TB/s HBM peak, i.e. launch gaps and everything else included; `kernel` names the
                 phase kernel with the LOWER per-kernel fraction and `per_kernel` lists both, from
                 HIP events recorded on the solver's stream during the timed region;
  cpu_baseline : the UNMODIFIED reference (oracle/_ref, built from /root/reference by
                 oracle/Makefile) — or our C port if that .so is absent — timed on this
                 box's host cores on a bounded sample of the same workload, 1 thread;
  cpu_baseline_all_cores : the same library called from one host thread per core on independent
                 planes — the reference's file-level OpenMP parallelism (jpeg2png.c:330; its
                 in-solver OpenMP gains nothing for a single plane).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_GRADIENT = 16              # per canvas pixel per launch (phase A)
BYTES_PROJECT = 22               # phase B
BYTES_ITERATION = BYTES_GRADIENT + BYTES_PROJECT
WEIGHT, PWEIGHT = 0.3, 0.001     # jpeg2png.c:22-23 defaults
EVENT_PAIR_US = None             # what a bracket of two HIP event records costs by itself (calibrated by the solver)
RCCL_LEG_TIMEOUT_S = 180         # watchdog of the Python RCCL harness leg of an N > 1 run
C_LEG_TIMEOUT_S = 120            # ... and of each child process that runs one leg of the C row tiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["headline", "batch"], default="headline")
    ap.add_argument("--size", type=int, default=0, help="override plane width (debug)")
    ap.add_argument("--height", type=int, default=0, help="override plane height (debug, single GPU)")
    ap.add_argument("--iterations", type=int, default=0, help="override iterations per solve (debug)")
    ap.add_argument("--batch", type=int, default=32, help="--config batch: images per step")
    ap.add_argument("--slots", type=int, default=8, help="--config batch: images in flight per GPU (measured 4 / 6 / 8 / 12: 188 / 197 / 203 / 193 images/s, profiles/r03_batch_slots.jsonl)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-host-to-host", action="store_true")
    ap.add_argument("--timing-every", type=int, default=16, help="HIP-event sample stride (iterations)")
    ap.add_argument("--force-tiled", action="store_true", help="run the row-tiled path even with one rank (debug)")
    ap.add_argument("--bands", type=int, default=0, help="--force-tiled on one GPU: number of bands on device 0")
    ap.add_argument("--tiled-impl", choices=["both", "c", "rccl"], default="both",
                    help="N > 1: which row-tiling engine(s) to time (default both; the faster one is `value`)")
    ap.add_argument("--norm-fold", type=int, default=-1, help="A/B: 1 = norm reduction inside k_gradient, 0 = stand-alone kernels; default: the library's choice")
    ap.add_argument("--nt", type=int, default=-1, help="A/B: non-temporal level 0..3 of the phase kernels (J2P_OPT_NT_GRADIENT); default: the library's policy")
    ap.add_argument("--narrow", type=int, default=-1, help="A/B: 0 = the projection reads int16 coefficients although one byte each would do (J2P_OPT_NARROW_COEFFICIENTS)")
    ap.add_argument("--norm-in-project", type=int, default=-1, help="A/B: final norm tree inside k_project (needs --norm-fold 1)")
    return ap.parse_args()


def _flush_c_stdio():
    """RCCL prints its NCCL_DEBUG=VERSION banner through C stdio, which is block-buffered on a pipe and would
    otherwise surface at exit, after the JSON line: push it out when the communicators exist"""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(width, seed):
    """time the reference solver on a bounded sample of the same workload: the same plane
    (same seed, Y-only, Q10, same weights), 4096 rows x `width`, but 40 iterations instead of
    500 (per-iteration cost is constant), 1 thread (1-channel/joint mode gains nothing from
    OpenMP)."""
    from jpeg2png_amd import synth
    from oracle import bindings as ob
    rows, its = 4096, 40
    if width > 4096:                       # the 16384-wide workload of an N > 1 run: the same number of pixels
        rows = 1024
    planes = synth.make_planes(width, rows, "444", 10, seed=seed, y_only=True)
    for p in planes:
        p.fdata = ob.decode_plane(p)
    if ob.have_ref():
        _, _, secs = ob.ref_compute(planes, WEIGHT, [PWEIGHT], its)
        kind = "reference"
    else:
        t0 = time.perf_counter()
        ob.oracle_compute(planes, WEIGHT, [PWEIGHT], its)
        secs = time.perf_counter() - t0
        kind = "port"
    one = {"value": round(width * rows * its / secs / 1e6, 2), "unit": "Mpixel-iterations/s", "cores": 1,
           "kind": kind, "cpu": cpu_model(), "width": width, "rows": rows, "iterations": its, "seconds": round(secs, 3),
           "planes": 1,
           "sample": f"{width}x{rows} Y-only Q10, {its} iterations, weight {WEIGHT}, pweight {PWEIGHT}, "
                     f"{secs:.2f} s inside compute(), host has {os.cpu_count()} cores"}
    # all cores: one compute() per host thread on independent planes (the reference's omp-parallel-for over files,
    # jpeg2png.c:330): the plane cut into 512-row pieces, each thread solves one piece, 10 iterations
    ncores = os.cpu_count() or 1
    nthreads = min(ncores, 256)
    piece_rows, its_all = 512 * 4096 // width if width > 4096 else 512, 10
    pieces = []
    for k in range(8):
        pl = synth.make_planes(width, rows, "444", 10, seed=seed, y_only=True,
                               rows=((k * piece_rows) % rows, (k * piece_rows) % rows + piece_rows))
        for p in pl:
            p.fdata = ob.decode_plane(p)
        pieces.append(pl)
    fn = (lambda pl: ob.ref_compute(pl, WEIGHT, [PWEIGHT], its_all)) if kind == "reference" else \
         (lambda pl: ob.oracle_compute(pl, WEIGHT, [PWEIGHT], its_all))
    threads = [threading.Thread(target=fn, args=(pieces[i % 8],)) for i in range(nthreads)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    secs_all = time.perf_counter() - t0
    allc = {"value": round(nthreads * width * piece_rows * its_all / secs_all / 1e6, 2), "unit": "Mpixel-iterations/s",
            "cores": nthreads, "kind": kind, "cpu": cpu_model(), "width": width, "rows": piece_rows, "iterations": its_all,
            "seconds": round(secs_all, 3), "planes": nthreads,
            "sample": f"{nthreads} concurrent compute() calls (one host thread each, the reference's file-level "
                      f"parallelism), each {width}x{piece_rows} Y-only Q10, {its_all} iterations; {secs_all:.2f} s wall"}
    return one, allc


def _timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def pick_leg(timed, truth):
    """Which leg of the row-tiled bench is `value`: the fastest among the legs whose plane hashes like `truth` (the same
    canvas solved whole on one GPU); without a truth, or for a leg that left no hash (the per-rank Python harness),
    `verified` is None and such legs only count when no leg could be verified; a leg with a WRONG plane never counts
    while any other leg exists.  Sets leg["verified"]; returns the name."""
    for v in timed.values():
        v["verified"] = None if (truth is None or v.get("digest") is None) else v["digest"] == truth
    good = ({k: v for k, v in timed.items() if v["verified"]} or {k: v for k, v in timed.items() if v["verified"] is None} or timed)
    return min(good, key=lambda k: good[k]["elapsed"])


def plane_digest(array):
    import hashlib
    return hashlib.blake2b(np.ascontiguousarray(array), digest_size=16).hexdigest()


BENCH_DIGESTS = os.path.join(ROOT, "tests", "golden", "bench_digests.json")


def reference_digest(W, H, its, seed):
    """(entry name, blake2b-128) of the plane the UNMODIFIED reference returns (compute.c:455-461) for this bench workload,
    from tests/golden/bench_digests.json (generated on the CPU box by tests/golden/make_golden.py --bench-digests from
    oracle/_ref; tests/test_bench_logic.py keeps the file honest) — or (None, None) for a workload without an entry"""
    try:
        with open(BENCH_DIGESTS) as f:
            entries = json.load(f)
    except (OSError, ValueError):
        return None, None
    for name, e in entries.items():
        if (isinstance(e, dict) and "rows_of_the_image" not in e and (e.get("W"), e.get("H"), e.get("iterations"), e.get("seed")) == (W, H, its, seed)
                and e.get("quality") == 10 and abs(e.get("weight", -1) - WEIGHT) < 1e-12 and abs(e.get("pweight", -1) - PWEIGHT) < 1e-12):
            return name, e["digest"]
    return None, None


def parity_object(fingerprint, W, H, its, seed):
    """the second half of BASELINE.json's metric ("...; per-plane PSNR vs CPU reference") for the plane the timed solver
    left behind: bit-identical to the reference's plane (infinite PSNR), or not, or no digest on file for this workload"""
    name, want = reference_digest(W, H, its, seed)
    return {"vs": "the unmodified reference's plane for this workload, by digest (tests/golden/bench_digests.json, blake2b-128)",
            "entry": name, "bit_identical": None if want is None else fingerprint == want,
            "psnr_db": "inf" if (want is not None and fingerprint == want) else None,
            "digest": fingerprint, "reference_digest": want}


def whole_canvas_on_one_gpu(j, plane, its, device=0, reps=2, digest=False):
    """a Y plane solved whole on ONE GPU, resident (reset + run): Mpx-it/s and the iteration fraction"""
    s = j.Solver([plane], WEIGHT, [PWEIGHT], its, device=device)

    def run():
        s.reset()
        s.run(its)
        s.sync()
    dt = _timed(run, reps)
    fingerprint = plane_digest(s.download(0)) if digest else None
    s.close()
    j.load_library().j2p_pool_trim()
    px = plane.w * plane.h
    out = {"ms_per_solve": round(dt * 1e3, 3), "us_per_iteration": round(dt / its * 1e6, 2),
           "Mpx_it_per_s": round(px * its / dt / 1e6, 1),
           "iteration_frac": round(BYTES_ITERATION * px * its / dt / 1e9 / HBM_PEAK_GBS, 4)}
    if digest:
        out["digest"] = fingerprint
    return out


def batch_images_per_s(j, synth, devices, n_images, slots, its=100):
    """configs[4]: n_images x 1080p 4:2:0 Q50 -i `its` joint through the C batch engine over `devices` (host
    coefficient buffers in, RGB out: PCIe inclusive; the file loop jpeg2png.c:330-337)"""
    planes = synth.make_planes(1920, 1080, "420", 50, seed=1238)
    # the RGB outputs go into a ring of arrays that exist before the clock starts and are reused between jobs, as a
    # pipeline that writes one image out while the next ones are solved would: mapping or first-touching host memory
    # while other jobs' kernels run stalls a launch of theirs each time — 196-201 images/s with an
    # array allocated per job against 229 this way on one GPU (profiles/r05_batch_prealloc.jsonl)
    ring = [np.zeros((1080, 1920, 3), np.uint8) for _ in range(4 * len(devices) * slots)]
    with j.Batch(devices=devices, slots_per_device=slots) as b:
        def step(n):
            pending = []
            for i in range(n):
                if len(pending) >= len(ring):
                    b.wait(pending.pop(0))
                pending.append(b.submit(planes, WEIGHT, [PWEIGHT] * 3, its, width=1920, height=1080, bits=8, out=ring[i % len(ring)]))
            for t in pending:
                b.wait(t)
        step(min(n_images, 4 * len(devices) * slots))          # warm the pool on every device
        t0 = time.perf_counter()
        step(n_images)
        dt = time.perf_counter() - t0
    W, H = 1920, 1088
    return {"images": n_images, "seconds": round(dt, 4), "images_per_s": round(n_images / dt, 2),
            "Mpx_it_per_s": round(n_images * W * H * 3 * its / dt / 1e6, 1), "devices": len(devices), "slots_per_device": slots}


def other_configs(j, synth):
    """configs[0], configs[1], a configs[4] slice and the N = 1 point of the row-tiled workload on this GPU,
    solver-resident like the headline (reset + run)."""
    out = []
    planes = synth.make_planes(512, 512, "420", 10, seed=1235)
    s = j.Solver(planes, WEIGHT, [PWEIGHT] * 3, 50)

    def run0():
        s.reset()
        s.run(50)
        s.sync()
    dt = _timed(run0, 20)
    s.close()
    out.append({"config": "configs[0] 512x512 4:2:0 Q10 -i 50 joint", "ms_per_solve": round(dt * 1e3, 4),
                "Mpx_it_per_s": round(512 * 512 * 3 * 50 / dt / 1e6, 1)})

    planes = synth.make_planes(1920, 1080, "444", 10, seed=1236)
    solvers = [j.Solver([p], WEIGHT if c == 0 else 0.0, [PWEIGHT], 100) for c, p in enumerate(planes)]

    def run1():
        for sv in solvers:
            sv.reset()
        # 10-iteration slices round-robin from one host thread: the three streams stay fed without three GIL-bound threads
        for _ in range(10):
            for sv in solvers:
                sv.run(10)
        for sv in solvers:
            sv.sync()
    dt = _timed(run1, 5)
    for sv in solvers:
        sv.close()
    out.append({"config": "configs[1] 1920x1080 4:4:4 Q10 -i 100, -s: three compute(1,...) on three streams, weights 0.3/0/0",
                "ms_per_image": round(dt * 1e3, 4), "Mpx_it_per_s": round(1920 * 1080 * 3 * 100 / dt / 1e6, 1)})

    planes = synth.make_planes(1920, 1080, "420", 50, seed=1238)
    n = 8
    solvers = [j.Solver(planes, WEIGHT, [PWEIGHT] * 3, 100) for _ in range(n)]
    W, H = solvers[0].W, solvers[0].H

    def run4():
        for sv in solvers:
            sv.reset()
        for _ in range(10):
            for sv in solvers:
                sv.run(10)
        for sv in solvers:
            sv.sync()
    dt = _timed(run4, 2)
    for sv in solvers:
        sv.close()
    out.append({"config": f"configs[4] slice: {n} x 1080p 4:2:0 Q50 -i 100 joint (canvas {W}x{H}), one stream each, resident",
                "ms_per_batch": round(dt * 1e3, 3), "images_per_s": round(n / dt, 2),
                "Mpx_it_per_s": round(n * W * H * 3 * 100 / dt / 1e6, 1)})
    j.load_library().j2p_pool_trim()

    # the N = 1 point of `--gpus N` (2048 rows of the 16384-wide plane per GPU): one such band, whole, on this GPU
    band = synth.make_planes(16384, 2048, "444", 10, seed=1234 + 4, y_only=True)[0]
    r = whole_canvas_on_one_gpu(j, band, 100, digest=True)
    r["parity"] = parity_object(r.pop("digest"), 16384, 2048, 100, 1234 + 4)
    r["config"] = ("configs[3] N = 1 point: 16384x2048 Y-only Q10 -i 100 on one GPU (what `--gpus N` gives every GPU: "
                   "2048 rows of the 16384-wide plane)")
    out.append(r)
    return out


def host_to_host(j, planes, its, resident_ms, device=0, reps=5):
    """SURVEY.md §8(d) "with and without H2D/D2H + aux_init": the C drop-in j2p_compute() — what compute() (compute.h:8)
    is — called like the reference's decode_file() calls it: libc-allocated pageable planes in (int16 coefficients + the
    decoded float plane), the float canvas plane back in newly allocated memory (compute.c:278-310, 455-461 are inside
    the reference's compute() too).  Never `value`."""
    for p in planes:
        if p.fdata is None:
            p.fdata = j.decode_plane(p, device=device)
    splits = []
    _, secs = j.compute_c(planes, WEIGHT, [PWEIGHT] * len(planes), its, device=device, repeat=reps + 1, splits=splits)
    in_order = [round(s * 1e3, 3) for s in secs]              # call 0 creates the arena (pool miss): listed, not counted
    ms = sorted(s * 1e3 for s in secs[1:])
    split_ms = {}
    for key in ("create", "issue", "housekeeping", "wait", "download", "destroy"):
        vals = sorted(sp[key + "_ms"] for sp in splits[1:])
        if vals:
            split_ms[key] = {"median": round(vals[len(vals) // 2], 3), "max": round(vals[-1], 3)}
    px = sum(p.w * p.w_samp * p.h * p.h_samp for p in planes[:1]) * len(planes)
    up = sum(p.w * p.h * 6 for p in planes)
    down = px * 4
    med = ms[len(ms) // 2]
    create_in_order = [round(sp["create_ms"], 3) for sp in splits]
    return {"ms_per_call": round(med, 3), "ms_per_call_all": [round(x, 3) for x in ms], "resident_ms_per_solve": round(resident_ms, 3),
            "boundary_ms": round(med - resident_ms, 3), "upload_bytes": up, "download_bytes": down,
            "ms_per_call_in_call_order": in_order, "create_ms_in_call_order": create_in_order,
            "first_call_note": "call 0 (first in the lists in call order) finds no arena in the pool and is left out of ms_per_call, "
                               "ms_per_call_all and split_ms; create is the pageable upload of upload_bytes + aux_init, i.e. the host's "
                               "pageable-copy rate: " + (f"{up / (split_ms['create']['median'] * 1e-3) / 1e9:.1f} GB/s at the median" if split_ms.get("create") else "n/a"),
            "split_ms": split_ms,
            "split_about": "j2p_compute_timing() per call, median and max over the calls: create = upload + aux_init (compute.c:278-310) with the "
                           "output planes allocated and touched beside it on helper threads (housekeeping: how long that took; create ends when "
                           "both are done), issue = queueing the loop, wait = until the last iteration has finished, download (compute.c:455-461), "
                           "destroy = freeing the inputs + the solver.  Nothing maps, unmaps or faults host memory while the loop runs: doing so "
                           "stalls one launch of the loop 12-16 ms (profiles/r05_host_to_host.jsonl)",
            "Mpx_it_per_s": round(px * its / (med * 1e-3) / 1e6, 1),
            "what": "j2p_compute() (the C drop-in behind compute(), compute.h:8) from libc-allocated pageable planes: upload of the "
                    "int16 coefficients and the decoded float plane, aux_init, all iterations, download into a newly allocated "
                    "plane; the output pages are touched beside the upload, the whole iteration loop is queued at once, the input planes "
                    "are freed behind the download (compute_host.c)"}


def bench_batch(a, j, synth):
    """configs[4] slice through the C batch API: host buffers in (int16 coefficients), RGB bytes out"""
    its = a.iterations or 100
    planes = synth.make_planes(1920, 1080, "420", 50, seed=1238)
    ndev = max(1, a.gpus)
    ring = [np.zeros((1080, 1920, 3), np.uint8) for _ in range(4 * ndev * a.slots)]      # (see batch_images_per_s)
    with j.Batch(devices=list(range(ndev)), slots_per_device=a.slots) as b:
        def step():
            pending = []
            for i in range(a.batch):
                if len(pending) >= len(ring):
                    b.wait(pending.pop(0))
                pending.append(b.submit(planes, WEIGHT, [PWEIGHT] * 3, its, width=1920, height=1080, bits=8, out=ring[i % len(ring)]))
            for t in pending:
                b.wait(t)
        for _ in range(a.warmup):
            step()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        dt = (time.perf_counter() - t0) / a.steps
    W, H = 1920, 1088
    print(json.dumps({
        "metric": "Mpixel-iterations/sec, batch of 1080p 4:2:0 images (host buffers in, RGB out)",
        "value": round(a.batch * W * H * 3 * its / dt / 1e6, 1), "unit": "Mpixel-iterations/s", "n_gpus": ndev,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "images_per_s": round(a.batch / dt, 2),
        "config": {"workload": f"{a.batch} x 1080p 4:2:0 Q50 -i {its} joint per step (BASELINE configs[4] slice), "
                               f"j2p_batch: {ndev} device(s) x {a.slots} slots, PCIe inclusive"}}), flush=True)


class Ranks:
    """the launch's ranks: RCCL (backend nccl) process group as the contract asks, plus a gloo side group for
    everything that must not touch the GPUs — the timing barriers (an RCCL barrier is a kernel that spins on the
    waiting rank's GPU: rank 0's band kernels of the C engine run on those very GPUs) and the max-over-ranks"""

    def __init__(self, rank, world, local_rank, one_device):
        self.rank, self.world = rank, world
        self.dist = None
        self.cpu = None
        if world > 1:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            if one_device:
                # rehearsing the multi-rank control flow on a 1-GPU box: RCCL refuses two ranks on one device
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            self.dist = dist
            self.cpu = dist.new_group(backend="gloo")

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier(group=self.cpu)

    def max(self, v):
        if self.dist is None:
            return v
        import torch
        t = torch.tensor([v], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.cpu)
        return float(t.item())

    def share(self, obj, src=0):
        if self.dist is None:
            return obj
        box = [obj]
        self.dist.broadcast_object_list(box, src=src, group=self.cpu)
        return box[0]

    def close(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()


def time_steps(ranks, reset, solve, sync, warmup, steps, eng, timing_every):
    """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides; max over ranks"""
    import torch
    for _ in range(warmup):
        reset()
        solve()
    sync()
    if eng is not None:
        eng.enable_timing(timing_every)
    ranks.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        reset()
        solve()
    sync()
    torch.cuda.synchronize()
    ranks.barrier()
    elapsed = ranks.max(time.perf_counter() - t0)
    g_ms, p_ms, samples = eng.kernel_times() if eng is not None else (0.0, 0.0, 0)
    if eng is not None:
        global EVENT_PAIR_US
        EVENT_PAIR_US = round(eng.timing_overhead_ms() * 1e3, 3)
        eng.enable_timing(0)
    return elapsed, g_ms, p_ms, samples


def per_kernel_roofline(px_gradient, px_project, g_ms, p_ms):
    per_kernel = {}
    for kname, kms, kb, px in (("k_gradient", g_ms, BYTES_GRADIENT, px_gradient), ("k_project", p_ms, BYTES_PROJECT, px_project)):
        gbs = px * kb / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        per_kernel[kname] = {"avg_launch_ms": round(kms, 4), "algorithmic_bytes_per_launch": px * kb,
                             "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
    return per_kernel


def _pmc_phase_bytes(path):
    """k_gradient + k_project bytes per launch at the L2's memory side from one tools/pmc_summary.py file, or None"""
    try:
        with open(path) as f:
            summ = json.load(f)
    except (OSError, ValueError):
        return None
    tot = {}
    for name, v in summ.items():
        if isinstance(v, dict) and "hbm_bytes_per_launch" in v:
            for kk in ("k_gradient", "k_project"):
                if name.startswith("j2p::" + kk) or name.startswith(kk):
                    tot[kk] = v["hbm_bytes_per_launch"]
    return tot["k_gradient"] + tot["k_project"] if len(tot) == 2 else None


def pmc_traffic(W=4096, H=4096):
    """HBM bytes per iteration of a WxH Y plane solved on one GPU, from the committed rocprofv3 PMC passes (profiles/,
    corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE x2 + WRITE_SIZE): the per-shape files of round 6
    (profiles/r06_pmc_<W>x<H>.json: 4096^2, the band shapes and the canvases past the Infinity Cache), else — the
    headline shape only — the newest rNN_pmc_summary.json"""
    shape = os.path.join(ROOT, "profiles", f"r06_pmc_{W}x{H}.json")
    got = _pmc_phase_bytes(shape)
    if got is not None:
        return got, f"profiles/r06_pmc_{W}x{H}.json (rocprofv3 --pmc, FETCH_SIZE x2 + WRITE_SIZE; k_gradient + k_project of a {W}x{H} plane on one GPU)"
    if (W, H) == (4096, 4096):
        for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
            got = _pmc_phase_bytes(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"))
            if got is not None:
                return got, f"profiles/{tag}_pmc_summary.json (FETCH_SIZE x2 + WRITE_SIZE; k_gradient + k_project)"
    return None, None


def band_pmc_traffic(W, rows):
    """HBM bytes per iteration of ONE band of a row-tiled run (k_gradient + k_project of a `rows`-row band of a W-wide
    plane, exchange `direct`), from profiles/r06_pmc_band.json — collected on one GPU with two bands (tools/band_pmc.py)"""
    path = os.path.join(ROOT, "profiles", "r06_pmc_band.json")
    try:
        with open(path) as f:
            about = json.load(f).get("_shape")
    except (OSError, ValueError):
        return None, None
    if about != [W, rows]:
        return None, None
    got = _pmc_phase_bytes(path)
    if got is None:
        return None, None
    return got, (f"profiles/r06_pmc_band.json (rocprofv3 --pmc, FETCH_SIZE x2 + WRITE_SIZE; k_gradient + k_project of one {W}x{rows} band "
                 "of a two-band run on one GPU, exchange direct: per GPU and iteration)")


def rocprof_kernel_us():
    """the phase kernels' average launch durations on the headline workload as rocprofv3 saw them (the committed kernel
    stats of the bench command, profiles/rNN_bench_kernel_stats.csv): what the HIP-event figures of this run stand beside"""
    import csv
    for tag in ("r06", "r05", "r04", "r03"):
        path = os.path.join(ROOT, "profiles", f"{tag}_bench_kernel_stats.csv")
        if not os.path.exists(path):
            continue
        out = {}
        for r in csv.DictReader(open(path)):
            for kk in ("k_gradient", "k_project", "k_norm_whole"):
                if ("j2p::" + kk) in r["Name"] and kk not in out:
                    out[kk] = round(float(r["AverageNs"]) / 1e3, 2)
        if "k_gradient" in out and "k_project" in out:
            return out, f"profiles/{tag}_bench_kernel_stats.csv"
    return None, None


def roofline_object(value_mpx, gpus_used, its, elapsed, steps, band_px, per_kernel, samples, timing_every, traffic=None, traffic_src=None):
    kern = min(per_kernel, key=lambda k: per_kernel[k]["frac"])
    it_gbs = BYTES_ITERATION * value_mpx * 1e6 / gpus_used / 1e9
    return {"bound": "hbm", "scope": "whole iteration (k_gradient + k_project, launch gaps included), wall clock, per GPU; "
                                     "`frac` is that; `kernel` names the phase kernel with the lower per-kernel fraction and "
                                     "`event_kernel_frac` is ITS fraction from this run's HIP-event brackets",
            "kernel": kern, "achieved": round(it_gbs, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(it_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
            "traffic_unit": "bytes per iteration", "traffic_source": traffic_src,
            "algorithmic_bytes_per_iteration": band_px * BYTES_ITERATION,
            "iteration_ms": round(elapsed / steps / its * 1e3, 5),
            "event_kernel_frac": per_kernel[kern]["frac"],
            "per_kernel": per_kernel,
            "event_samples": samples, "event_pair_overhead_us": EVENT_PAIR_US,
            "note": "per-kernel durations come from HIP events around every "
                    f"{timing_every}th iteration; the brackets themselves are in the figures (event_pair_overhead_us = what an EMPTY "
                    "bracket measures on the same stream: the scale of it; not subtracted, because with a kernel in between part of it "
                    "overlaps), so per-kernel durations read 2-3 us above rocprofv3's (profiles/), which are the reproducible ones",
            "what_limits_it": "k_project moves its bytes at the rate anything reaches through this part's fabric (6.1-6.3 TB/s: a float4 copy, "
                              "and k_project itself at 8192^2 where every byte comes from HBM — the same 58-60 us per 16.8 Mpixel as at 4096^2 "
                              "where most come from the Infinity Cache; peak above is the 8 TB/s spec the contract asks for); k_gradient is "
                              "bound by instruction issue: four resident wavefronts finish one 16-row strip per ~6.1 us per SIMD whatever "
                              "their order (oldest-first arbitration, profiles/r06_wave_trace.jsonl), 8.25 strips per SIMD, at the 1.96 GHz "
                              "the SIMDs sustain under it — and moves 1.15 x its algorithmic bytes (halo rows) meanwhile (DESIGN.md sections 4-5)"}


def single_gpu(a, j, synth, local_rank):
    W = a.size or 4096
    H = a.height or W
    its = a.iterations or 500
    seed = 1234 + 3
    workload = f"{W}x{H} Y-only Q10 -i {its} (BASELINE configs[2])"
    planes = synth.make_planes(W, H, "444", 10, seed=seed, y_only=True)
    solver = j.Solver(planes, WEIGHT, [PWEIGHT], its, device=local_rank)   # fdata=None: decoded on device
    if a.norm_fold >= 0:
        solver.debug_option(j.J2P_OPT_NORM_FOLD, a.norm_fold)
    if a.norm_in_project >= 0:
        solver.debug_option(j.J2P_OPT_NORM_IN_PROJECT, a.norm_in_project)
    if a.nt >= 0:
        solver.debug_option(j.J2P_OPT_NT_GRADIENT, a.nt)
    if a.narrow >= 0:
        solver.debug_option(j.J2P_OPT_NARROW_COEFFICIENTS, a.narrow)
    del planes
    ranks = Ranks(0, 1, local_rank, False)
    launches = solver.launches_per_iteration()
    elapsed, g_ms, p_ms, samples = time_steps(ranks, solver.reset, lambda: solver.run(its), solver.sync, a.warmup, a.steps,
                                              solver, a.timing_every)
    px = W * H
    value = px * its * a.steps / elapsed / 1e6
    # the plane the LAST timed step left behind (every step is a whole solve from iteration 0), against the reference's
    parity = parity_object(plane_digest(solver.download(0)), W, H, its, seed)
    per_kernel = per_kernel_roofline(px, px, g_ms, p_ms)
    traffic, traffic_src = pmc_traffic(W, H)          # (a shape without committed counters: null)
    roof = roofline_object(value, 1, its, elapsed, a.steps, px, per_kernel, samples, a.timing_every, traffic, traffic_src)
    if not a.size:
        # the reproducible per-kernel figures: rocprofv3's, from the committed stats of this same command
        prof, prof_src = rocprof_kernel_us()
        if prof:
            roof["rocprof_per_kernel"] = {
                "source": prof_src, "avg_launch_us": prof,
                "frac": {k: round(px * b / (prof[k] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                         for k, b in (("k_gradient", BYTES_GRADIENT), ("k_project", BYTES_PROJECT)) if k in prof}}
    out = {
        "metric": "Mpixel-iterations/sec on 4K Y-plane", "value": round(value, 1),
        "unit": "Mpixel-iterations/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "iterations_per_step": its, "weight": WEIGHT, "pweight": PWEIGHT,
                   "parallelism": "single GPU", "launches_per_iteration": launches},
        "roofline": roof,
        "parity": parity,
    }
    if parity["bit_identical"] is False:
        # a plane that is not the reference's: the figure is set aside (the driver treats a null value as unmeasured)
        out["unverified_value"] = out["value"]
        out["value"] = None
    solver.close()
    if not a.no_host_to_host and not a.size:
        try:
            planes = synth.make_planes(W, H, "444", 10, seed=seed, y_only=True)
            out["host_to_host"] = host_to_host(j, planes, its, elapsed / a.steps * 1e3, device=local_rank)
            del planes
        except Exception as e:          # noqa: BLE001
            out["host_to_host"] = {"error": f"{type(e).__name__}: {e}"}
    if not a.no_other_configs and not a.size:
        out["other_configs"] = other_configs(j, synth)
    if not a.no_cpu_baseline:
        out["cpu_baseline"], out["cpu_baseline_all_cores"] = cpu_baseline(W, seed)
    _flush_c_stdio()
    print(json.dumps(out), flush=True)


def tiled(a, j, synth, rank, world, local_rank, one_device):
    """N > 1 (or --force-tiled): the 16384-wide plane, 2048 rows per GPU, both engines"""
    import torch
    n_gpus = world
    ranks = Ranks(rank, world, local_rank, one_device)
    W = a.size or 16384
    rows_per_gpu = 2048 if not a.size else max(64, a.size // 8 // 16 * 16)
    nband = n_gpus if n_gpus > 1 else max(2, a.bands or 2)
    H = rows_per_gpu * nband
    its = a.iterations or 100
    seed = 1234 + 4
    workload = (f"{W}x{H} Y-only Q10 -i {its}, row-tiled {rows_per_gpu} rows/GPU over {n_gpus} GPUs "
                f"(BASELINE configs[3] at 8 GPUs)")
    px, band_px = W * H, W * rows_per_gpu
    # every rank synthesises its own band(s); rank 0 collects them through /dev/shm for the engines it drives alone
    tag = f"/dev/shm/j2p_bench_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
    my_bands = {}
    whole_plane = None
    if n_gpus == 1:
        # --force-tiled on one GPU: the whole canvas, synthesised band by band on the host's cores
        whole_plane = synth.make_y_plane_banded(W, H, 10, seed, band_rows=rows_per_gpu, workers=min(nband, os.cpu_count() or 1))
        my_bands[0] = synth.Plane(W, rows_per_gpu, 1, 1, whole_plane.data[: W * rows_per_gpu], whole_plane.quant_table)
    else:
        band = synth.make_planes(W, H, "444", 10, seed=seed, y_only=True, rows=(rank * rows_per_gpu, (rank + 1) * rows_per_gpu))[0]
        my_bands[rank] = band
        np.save(f"{tag}_{rank}.npy", band.data)
        ranks.barrier()
        if rank == 0:
            whole_plane = synth.Plane(W, H, 1, 1, np.concatenate([np.load(f"{tag}_{b}.npy") for b in range(nband)]), band.quant_table)
        ranks.barrier()
        try:
            os.unlink(f"{tag}_{rank}.npy")
        except OSError:
            pass

    legs = {}
    want_c = a.tiled_impl in ("both", "c")
    # what this run can cost at worst: every child leg has its own timeout, the per-rank harness a watchdog (rank 0, stderr,
    # and in the line: the driver's window has to hold it)
    n_c_legs = (4 if nband > 2 else 2) + 2 if want_c else 0
    worst_case_s = n_c_legs * C_LEG_TIMEOUT_S + RCCL_LEG_TIMEOUT_S
    if rank == 0:
        print(f"bench: row-tiled run over {n_gpus} GPU(s): up to {n_c_legs} child legs of at most {C_LEG_TIMEOUT_S} s each + the per-rank "
              f"RCCL harness (watchdog {RCCL_LEG_TIMEOUT_S} s): {worst_case_s} s at worst before the whole-canvas solve, the batch and "
              "the CPU baseline", file=sys.stderr, flush=True)
    # (the RCCL harness needs one rank per GPU: not in the one-device rehearsal, and with a single rank only on request)
    want_rccl = (a.tiled_impl == "rccl" or (a.tiled_impl == "both" and world > 1)) and not (one_device and world > 1)

    # ---- the truth every leg's plane is compared with: the SAME canvas solved whole on ONE GPU (also the strong-scaling
    # denominator), by rank 0 ----
    whole = None
    if rank == 0 and not a.no_other_configs:
        try:
            whole = whole_canvas_on_one_gpu(j, whole_plane, its, device=local_rank, digest=True)
        except Exception as e:          # noqa: BLE001
            whole = {"error": f"{type(e).__name__}: {e}"}
    ranks.barrier()

    # ---- legs 1: the C engine in each of its exchanges (j2p_tiled.hip).  Rank 0 runs every leg in a CHILD process (this
    # file with --leg-child): none of the exchanges has met real multi-GPU hardware before the driver's run, and a leg
    # that hangs or dies there must cost its own figure, not the line.  The other ranks have nothing to do in these
    # legs (one process drives all GPUs) and wait in the CPU barrier; the child does its own W warm-up and K timed
    # steps between device synchronisations. ----
    plane_file = f"{tag}_whole.npy"
    if rank == 0 and want_c:
        np.save(plane_file, whole_plane.data)

    def c_leg(name, exchange=None, norm="root", wait=None):
        if rank == 0:
            devices = list(range(n_gpus)) if n_gpus > 1 and not one_device else [local_rank] * nband
            spec = {"plane": plane_file, "W": W, "H": H, "quant": [int(q) for q in np.asarray(whole_plane.quant_table).reshape(-1)],
                    "its": its, "devices": devices, "warmup": a.warmup, "steps": a.steps, "timing_every": a.timing_every}
            env = dict(os.environ)
            env.pop("J2P_TILED_EXCHANGE", None)
            env.pop("J2P_TILED_WAIT", None)
            if exchange:
                env["J2P_TILED_EXCHANGE"] = exchange                  # read by j2p_tiled_create
            else:
                env["J2P_TILED_VERIFY"] = "2"                         # the picker at work, and its measurements on stderr
            if wait:
                env["J2P_TILED_WAIT"] = wait
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
                env.pop(k, None)
            import subprocess
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg-child", json.dumps(spec)], env=env,
                                   capture_output=True, text=True, timeout=C_LEG_TIMEOUT_S)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("LEG ")]
                if r.returncode != 0 or not line:
                    raise RuntimeError(f"exit code {r.returncode}: " + (r.stderr.strip().splitlines() or ["no output"])[-1][:300])
                res = json.loads(line[-1][4:])
                if "error" in res:
                    raise RuntimeError(res["error"])
                how = res["exchange"]
                direct = ("row sums of g^2 pushed from k_gradient and edge rows from k_project as posted peer writes, ||g|| "
                          "reduced inside k_project: two launches per band and iteration; the projection waits for ")
                what = {"direct": direct + "the other bands' gradient events itself",
                        "direct, wait root": direct + "ONE event of band 0, which waited for every band's gradient event",
                        "direct, wait collector": direct + "ONE event of a helper stream that waited for the other bands' gradient events",
                        "direct, wait counter": direct + "ONE value in coherent host memory that every band's k_gradient counts up itself "
                                                         "(hipStreamWaitValue64; the neighbours' projections are awaited through hipStreamWriteValue64 flags): no HIP events",
                        "copy": "round 3's exchange: a copy kernel pulls the neighbours' edge rows, "
                                + ("one band reduces ||g|| for all" if norm == "root" else "every band reduces ||g|| itself")
                                + ": four launches per band and iteration",
                        "rccl": "ncclAllGather of the row sums + grouped ncclSend / ncclRecv of the edge rows on the band streams "
                                "(librccl dlopen()ed by the C library, one communicator per band)"}.get(how, how)
                # what the library said while it chose / verified its exchange on these GPUs: one line per candidate it
                # demoted or found unavailable, and — the picker's leg — each verified candidate's us per scratch iteration
                picker = [ln.strip() for ln in r.stderr.splitlines() if ln.startswith("jpeg2png_amd: row tiling")]
                legs[name] = {"elapsed": res["elapsed"], "g_ms": res["g_ms"], "p_ms": res["p_ms"], "samples": res["samples"], "split": False,
                              "host_cpu_s": res["host_cpu_s"], "digest": res["digest"], "exchange": how,
                              "create_s": res.get("create_s"), "picker_log": picker, "rccl": res.get("rccl"),
                              "parallelism": (f"row-tiled x{nband}: C engine (j2p_tiled), one process drives all GPUs, one host thread per "
                                              f"band; exchange '{how}': {what}")}
            except subprocess.TimeoutExpired:
                legs[name + "_error"] = f"no result within {C_LEG_TIMEOUT_S} s (child process killed)"
            except Exception as e:      # noqa: BLE001  (no peer access between the GPUs, no librccl, a fault on first contact ...)
                legs[name + "_error"] = f"{type(e).__name__}: {e}"
            if name + "_error" in legs:
                print(f"bench: C row tiling ({name}) gave no figure: {legs[name + '_error']}", file=sys.stderr, flush=True)
        ranks.barrier()

    if want_c:
        c_leg("c")
        c_leg("c_copy", "copy")
        if nband > 2:
            # (what a wait for ANOTHER GPU's event costs is unknown here — on one GPU 13 us each, profiles/r04_event_waits.json —
            # so the two other ways of learning that every band's gradient launch has finished are timed as well)
            c_leg("c_waitroot", "direct", wait="root")
            c_leg("c_waitcoll", "direct", wait="collector")

    # ---- leg 2: one process per GPU over RCCL ----
    watchdog = None
    state = {"printed": False, "out": None}
    lock = threading.Lock()

    def emit():
        with lock:
            if state["printed"] or state["out"] is None:
                return
            state["printed"] = True
            _flush_c_stdio()
            print(json.dumps(state["out"]), flush=True)

    def finish(extra_other=None, strict=True):
        """rank 0: assemble the JSON line from the legs measured so far"""
        if rank != 0:
            return
        timed = {k: v for k, v in legs.items() if isinstance(v, dict) and "elapsed" in v}
        if not timed and not strict:
            return
        if not timed:
            raise SystemExit("bench: no row-tiling engine could run: " + json.dumps({k: v for k, v in legs.items() if not isinstance(v, dict)}))
        # the truth: the whole-canvas solve's hash; without it (skipped / failed) the copy exchange's, round 3's engine
        truth = (whole or {}).get("digest") or timed.get("c_copy", {}).get("digest")
        best = pick_leg(timed, truth)
        L = timed[best]
        value = px * its * a.steps / L["elapsed"] / 1e6
        # pixels per TIMED launch: in the split schedules the events bracket the interior launches only
        # (all 16-row segments but the band's first and last; all block rows but the first and last)
        pxg, pxp = (W * (rows_per_gpu - 32), W * (rows_per_gpu - 16)) if L["split"] else (band_px, band_px)
        per_kernel = per_kernel_roofline(pxg, pxp, L["g_ms"], L["p_ms"])
        others = []
        for k, v in timed.items():
            if k == best:
                continue
            others.append({"config": f"the same workload through the other engine ({k})", "parallelism": v["parallelism"],
                           "Mpx_it_per_s": round(px * its * a.steps / v["elapsed"] / 1e6, 1),
                           "ms_per_step": round(v["elapsed"] / a.steps * 1e3, 3),
                           "us_per_iteration": round(v["elapsed"] / a.steps / its * 1e6, 2),
                           "bits_equal_to_the_whole_canvas_solve": v["verified"],
                           **{kk: v[kk] for kk in ("create_s", "picker_log", "rccl") if v.get(kk)}})
        for k, v in legs.items():
            if not isinstance(v, dict):
                others.append({"config": f"engine {k}", "error": v})
        others += extra_other or []
        gpus_used = n_gpus if n_gpus > 1 else 1
        # real bytes of ONE band's iteration (committed counters of a 2048-row band of the 16384-wide plane in a linked run)
        traffic, traffic_src = band_pmc_traffic(W, rows_per_gpu)
        out = {
            "metric": (f"Mpixel-iterations/sec on a 16384-wide Y plane row-tiled over {n_gpus} GPUs, 2048 rows per GPU "
                       "(BASELINE configs[3] at 8 GPUs)") if not a.size else "Mpixel-iterations/sec, row-tiled Y plane (debug size)",
            "value": round(value, 1),
            "unit": "Mpixel-iterations/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(L["elapsed"] / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "scaling_note": (f"weak: every GPU holds {rows_per_gpu} rows of the {W}-wide plane whatever N, so the canvas grows with N "
                             "and `value` is the whole job's rate on it; the ratio to ONE GPU solving that same canvas whole — a "
                             "strong-scaling figure — is reported separately in other_configs (strong_scaling_speedup)"),
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "iterations_per_step": its, "weight": WEIGHT, "pweight": PWEIGHT,
                       "parallelism": L["parallelism"], "engine": best, "band_threads_host_cpu_s": L.get("host_cpu_s"),
                       "bits_equal_to_the_whole_canvas_solve": L["verified"],
                       "us_per_iteration": round(L["elapsed"] / a.steps / its * 1e6, 2),
                       # the exchange was chosen by the library on these GPUs (leg `c`): what it verified, demoted, measured
                       "picker_log": (legs.get("c") or {}).get("picker_log") if isinstance(legs.get("c"), dict) else None,
                       "create_s": L.get("create_s"), "rccl": L.get("rccl"),
                       "worst_case_seconds_of_the_legs": worst_case_s},
            "roofline": roofline_object(value, gpus_used, its, L["elapsed"], a.steps, band_px, per_kernel, L["samples"], a.timing_every,
                                        traffic, traffic_src),
            "parity": parity_object(L.get("digest"), W, H, its, seed),
            "other_configs": others,
        }
        if out["parity"]["bit_identical"] is False or L["verified"] is False:
            out["unverified_value"] = out["value"]
            out["value"] = None
        state["out"] = out

    finish(strict=False)

    def bark(which):
        # an RCCL leg hung (its first meeting with real multi-GPU hardware is the driver's run) or a rank failed inside
        # it: rank 0 reports what the legs before it measured
        if rank == 0:
            if state["out"] is not None:
                state["out"]["other_configs"].append({"config": f"engine {which}", "error": f"no result within {RCCL_LEG_TIMEOUT_S} s"})
            emit()
        os._exit(0 if (rank != 0 or state["printed"]) else 1)

    # ---- leg 1c: the C engine over RCCL (librccl dlopen()ed by the library) ----
    if want_c and n_gpus > 1 and not one_device:
        c_leg("c_rccl", "rccl")
        finish(strict=False)
    elif want_c and rank == 0:
        legs["c_rccl_error"] = "not run: the rccl exchange needs one GPU per band (this is a one-GPU run)"
        finish(strict=False)
    if want_c:
        # ... and the one without events: the gradient kernels count up a value in host memory, one hipStreamWaitValue64
        # (bands on GPUs of their own; with more than two bands on one GPU the engine takes the event form by itself)
        if (n_gpus > 1 and not one_device) or nband <= 2:
            c_leg("c_counter", "direct", wait="counter")
        finish(strict=False)
    if rank == 0 and want_c:
        try:
            os.unlink(plane_file)
        except OSError:
            pass
    if want_rccl:
        watchdog = threading.Timer(RCCL_LEG_TIMEOUT_S, bark, args=("rccl",))
        watchdog.daemon = True
        watchdog.start()
        err = ""
        try:
            from jpeg2png_amd import tiled as tiledmod
            dist = ranks.dist
            if dist is None:
                import torch.distributed as dist
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29541")
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
            r0, r1 = rank * rows_per_gpu, (rank + 1) * rows_per_gpu
            bp = my_bands[rank]
            bp = synth.Plane(bp.w, H, 1, 1, bp.data, bp.quant_table)   # planes describe the whole image; arrays are band-local
            if n_gpus == 1:                                             # --force-tiled on one GPU: one band, its own neighbour
                bp = synth.Plane(bp.w, rows_per_gpu, 1, 1, bp.data, bp.quant_table)
                r0, r1 = 0, rows_per_gpu
            engine = tiledmod.HipBandEngine([bp], WEIGHT, [PWEIGHT], its, (r0, r1), local_rank)
            # (J2P_RCCL_OVERLAP=1: the split phases with the exchange on a side stream; default: one gradient and one
            # projection launch per iteration like the C engine — on a band with a GPU to itself the split costs more than
            # the exchange it hides, profiles/r03_band_alone.jsonl)
            driver = tiledmod.RowTiledSolver(engine, overlap=os.environ.get("J2P_RCCL_OVERLAP", "0") == "1")
            _flush_c_stdio()

            def reset():
                engine.reset()
                driver.start()
            reset()
            elapsed, g_ms, p_ms, samples = time_steps(ranks, reset, lambda: driver.iterate(its), engine.solver.sync, a.warmup,
                                                      a.steps, engine.solver, a.timing_every)
            # (a single rank solves ONE band of the nband-band canvas the figures below are quoted on)
            legs["rccl"] = {"elapsed": elapsed * (nband if n_gpus == 1 else 1), "g_ms": g_ms, "p_ms": p_ms, "samples": samples,
                            "split": bool(driver.overlap),
                            "parallelism": (f"row-tiled x{n_gpus}, one process per GPU, RCCL halo send/recv + norm all-gather "
                                            f"({'librccl called on the solver streams' if driver.direct is not None else 'through torch.distributed'})")}
            driver.close()
            engine.close()
        except Exception as e:          # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
            legs["rccl_error"] = err
            print(f"bench: rank {rank}: RCCL harness failed: {err}", file=sys.stderr, flush=True)
        ranks.barrier()                 # a rank that failed alone leaves the others in a collective: the watchdog ends that
        watchdog.cancel()
    finish()

    # ---- the same canvas on ONE GPU, configs[4] over all GPUs, the CPU baseline (rank 0) ----
    if rank == 0 and state["out"] is not None and not a.no_other_configs:
        extra = []
        if whole is not None and "error" not in whole:
            r = {k: v for k, v in whole.items() if k != "digest"}
            r["parity"] = parity_object(whole.get("digest"), W, H, its, seed)
            r["config"] = (f"strong-scaling denominator: the SAME {W}x{H} canvas solved whole on ONE GPU, -i {its} "
                           "(value / this = speed-up of the tiling; its plane's hash is what every leg is compared with)")
            r["strong_scaling_speedup"] = round((state["out"]["value"] or state["out"].get("unverified_value") or 0.0) / r["Mpx_it_per_s"], 3)
            extra.append(r)
        else:
            extra.append({"config": "strong-scaling denominator", "error": (whole or {}).get("error", "not run")})
        try:
            devs = list(range(n_gpus)) if n_gpus > 1 and not one_device else [local_rank]
            r = batch_images_per_s(j, synth, devs, 256 if not a.size else 16, a.slots)
            r["config"] = (f"configs[4]: {r['images']} x 1080p 4:2:0 Q50 -i 100 joint through j2p_batch over {len(devs)} GPU(s) "
                           "(host coefficient buffers in, RGB out: PCIe inclusive)")
            extra.append(r)
        except Exception as e:          # noqa: BLE001
            extra.append({"config": "configs[4] batch", "error": f"{type(e).__name__}: {e}"})
        state["out"]["other_configs"] += extra
    if rank == 0 and state["out"] is not None and not a.no_cpu_baseline:
        state["out"]["cpu_baseline"], state["out"]["cpu_baseline_all_cores"] = cpu_baseline(W, seed)
    ranks.barrier()
    if rank == 0:
        emit()
    ranks.close()


def leg_child(spec_json):
    """one leg of the C row tiling in a process of its own (see tiled()): create, W warm-up steps, K timed steps, the
    hash of the plane left behind; prints `LEG {json}`"""
    spec = json.loads(spec_json)
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    out = {}
    try:
        plane = synth.Plane(spec["W"], spec["H"], 1, 1, np.load(spec["plane"]), np.array(spec["quant"], dtype=np.uint16))
        its = spec["its"]
        t_create = time.perf_counter()
        with j.TiledSolver([plane], WEIGHT, [PWEIGHT], its, devices=spec["devices"]) as t:
            create_s = time.perf_counter() - t_create        # (the picker's verification of every candidate is in here)
            eng = t.band_solver(0)
            for _ in range(spec["warmup"]):
                t.reset()
                t.run(its)
            t.sync()
            eng.enable_timing(spec["timing_every"])
            t0 = time.perf_counter()
            for _ in range(spec["steps"]):
                t.reset()
                t.run(its)
            t.sync()
            elapsed = time.perf_counter() - t0
            g_ms, p_ms, samples = eng.kernel_times()
            eng.enable_timing(0)
            out = {"elapsed": elapsed, "g_ms": g_ms, "p_ms": p_ms, "samples": samples, "host_cpu_s": round(t.host_cpu_seconds(), 3),
                   "exchange": t.exchange(), "digest": plane_digest(t.download(0)), "create_s": round(create_s, 3)}
            if out["exchange"] == "rccl":
                out["rccl"] = {"ncclGetVersion": j.rccl_version(), "nranks_of_ncclCommInitAll": len(spec["devices"]),
                               "devices": spec["devices"]}
    except Exception as e:              # noqa: BLE001
        out = {"error": f"{type(e).__name__}: {e}"}
    _flush_c_stdio()
    print("LEG " + json.dumps(out), flush=True)


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--leg-child":
        leg_child(sys.argv[2])
        return
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    one_device = bool(os.environ.get("J2P_BENCH_ONE_DEVICE"))
    if one_device:                                  # rehearsal on a 1-GPU box: every rank on the same device
        local_rank = int(os.environ["J2P_BENCH_ONE_DEVICE"])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world == 1 and a.gpus > 1:
        raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")

    import torch
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    j.build()
    torch.cuda.set_device(local_rank)

    if a.config == "batch":
        if rank == 0:
            bench_batch(a, j, synth)
        return
    if world > 1 or a.force_tiled:
        tiled(a, j, synth, rank, world, local_rank, one_device)
    else:
        single_gpu(a, j, synth, local_rank)


if __name__ == "__main__":
    main()
