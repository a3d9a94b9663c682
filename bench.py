#!/usr/bin/env python3
"""bench.py — Mpixel-iterations/s of the jpeg2png deblocking solver on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`
prints ONE JSON line on rank 0.  A *step* is one complete solve of the workload —
reset to iteration 0 from inputs already resident in HBM, then all its iterations
(the step size radius/sqrt(1+iterations) ties the iterations of one solve together,
compute.c:443) — with no host copies inside the timed region.

Workloads (BASELINE.json configs):
  N = 1 : configs[2]  4096x4096 Y-only, Q=10, -i 500, weight 0.3, pweight 0.001
          (the configuration the metric "Mpixel-iterations/sec on 4K Y-plane" is quoted on)
  N > 1 : configs[3]  16384-wide Y-only plane, Q=10, -i 100, row-tiled: 2048 rows per GPU
          (N = 8 is exactly the 16384x16384 config); weak scaling (fixed rows per GPU).
          Default engine: the C row tiling (j2p_tiled: one process drives all N GPUs with one host
          thread per band; bands exchange edge rows and norm row sums by peer access over xGMI,
          ordered by HIP events) — rank 0 drives it, the other ranks of the launch only take part
          in the barriers.  `--tiled-impl rccl`: the round-1 harness, one process per GPU with
          RCCL send/recv + all-gather (jpeg2png_amd/tiled.py).
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
                 in-solver OpenMP gains nothing for a single plane);
  other_configs: configs[0], configs[1] and a configs[4] slice timed in this same run.
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
    ap.add_argument("--slots", type=int, default=4, help="--config batch: images in flight per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--timing-every", type=int, default=16, help="HIP-event sample stride (iterations)")
    ap.add_argument("--force-tiled", action="store_true", help="run the row-tiled path even with one rank (debug)")
    ap.add_argument("--bands", type=int, default=0, help="--force-tiled on one GPU: number of bands on device 0")
    ap.add_argument("--tiled-impl", choices=["c", "rccl"], default="c")
    ap.add_argument("--norm-fold", type=int, default=-1, help="A/B: 1 = norm reduction inside k_gradient, 0 = stand-alone kernels; default: the library's choice")
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
           "kind": kind, "cpu": cpu_model(),
           "sample": f"{width}x{rows} Y-only Q10, {its} iterations, weight {WEIGHT}, pweight {PWEIGHT}, "
                     f"{secs:.2f} s inside compute(), host has {os.cpu_count()} cores"}
    # all cores: one compute() per host thread on independent planes (the reference's omp-parallel-for over files,
    # jpeg2png.c:330): the 4096-row plane cut into 512-row pieces, each thread solves one piece, 20 iterations
    ncores = os.cpu_count() or 1
    nthreads = min(ncores, 256)
    piece_rows, its_all = 512, 10
    pieces = []
    for k in range(8):
        pl = synth.make_planes(width, 4096, "444", 10, seed=seed, y_only=True, rows=(k * piece_rows, (k + 1) * piece_rows))
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
            "cores": nthreads, "kind": kind, "cpu": cpu_model(),
            "sample": f"{nthreads} concurrent compute() calls (one host thread each, the reference's file-level "
                      f"parallelism), each {width}x{piece_rows} Y-only Q10, {its_all} iterations; {secs_all:.2f} s wall"}
    return one, allc


def other_configs(j, synth):
    """configs[0], configs[1] and a configs[4] slice on this GPU, solver-resident like the headline (reset + run)."""
    out = []

    def timed(fn, reps):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps

    planes = synth.make_planes(512, 512, "420", 10, seed=1235)
    s = j.Solver(planes, WEIGHT, [PWEIGHT] * 3, 50)

    def run0():
        s.reset()
        s.run(50)
        s.sync()
    dt = timed(run0, 20)
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
    dt = timed(run1, 5)
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
    dt = timed(run4, 2)
    for sv in solvers:
        sv.close()
    out.append({"config": f"configs[4] slice: {n} x 1080p 4:2:0 Q50 -i 100 joint (canvas {W}x{H}), one stream each, resident",
                "ms_per_batch": round(dt * 1e3, 3), "images_per_s": round(n / dt, 2),
                "Mpx_it_per_s": round(n * W * H * 3 * 100 / dt / 1e6, 1)})
    return out


def bench_batch(a, j, synth):
    """configs[4] slice through the C batch API: host buffers in (int16 coefficients), RGB bytes out"""
    its = a.iterations or 100
    planes = synth.make_planes(1920, 1080, "420", 50, seed=1238)
    ndev = max(1, a.gpus)
    with j.Batch(devices=list(range(ndev)), slots_per_device=a.slots) as b:
        def step():
            tickets = [b.submit(planes, WEIGHT, [PWEIGHT] * 3, its, width=1920, height=1080, bits=8) for _ in range(a.batch)]
            for t in tickets:
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


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    one_device = bool(os.environ.get("J2P_BENCH_ONE_DEVICE"))
    if one_device:                                  # debugging on a 1-GPU box: every rank on the same device
        local_rank = int(os.environ["J2P_BENCH_ONE_DEVICE"])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = a.gpus
    if world != n_gpus:
        if world == 1 and n_gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        n_gpus = world

    import torch
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    j.build()
    torch.cuda.set_device(local_rank)

    if a.config == "batch":
        if rank == 0:
            bench_batch(a, j, synth)
        return

    tiled_mode = n_gpus > 1 or a.force_tiled
    c_tiled = tiled_mode and a.tiled_impl == "c"
    driver = None
    dist = None
    if n_gpus > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            if one_device:
                # debugging the multi-rank control flow on a 1-GPU box: RCCL refuses two ranks on one device, gloo does not
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if not tiled_mode:
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
        del planes
        reset, solve, sync = solver.reset, (lambda: solver.run(its)), solver.sync
        eng = solver
        parallelism = "single GPU"
        rows_per_gpu = H
    elif c_tiled:
        W = a.size or 16384
        rows_per_gpu = 2048 if not a.size else max(64, a.size // 8 // 16 * 16)
        nband = n_gpus if n_gpus > 1 else max(2, a.bands or 2)
        H = rows_per_gpu * nband
        its = a.iterations or 100
        seed = 1234 + 4
        workload = (f"{W}x{H} Y-only Q10 -i {its}, row-tiled {rows_per_gpu} rows/GPU over {n_gpus} GPUs "
                    f"(BASELINE configs[3] at 8 GPUs)")
        # every rank synthesises its own band(s); rank 0 collects them through /dev/shm and drives all GPUs
        tag = f"/dev/shm/j2p_bench_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
        mine = range(nband) if n_gpus == 1 else [rank]
        for b in mine:
            band = synth.make_planes(W, H, "444", 10, seed=seed, y_only=True, rows=(b * rows_per_gpu, (b + 1) * rows_per_gpu))[0]
            np.save(f"{tag}_{b}.npy", band.data)
            qt = band.quant_table
        if dist is not None:
            dist.barrier()
        eng = None
        c_ok = [True, ""]
        if rank == 0:
            try:
                data = np.concatenate([np.load(f"{tag}_{b}.npy") for b in range(nband)])
                plane = synth.Plane(W, H, 1, 1, data, qt)
                devices = list(range(n_gpus)) if n_gpus > 1 and not one_device else [local_rank] * nband
                tsolver = j.TiledSolver([plane], WEIGHT, [PWEIGHT], its, devices=devices)
                del data, plane
                eng = tsolver.band_solver(0)
                reset, solve, sync = tsolver.reset, (lambda: tsolver.run(its)), tsolver.sync
            except Exception as e:      # noqa: BLE001  (no peer access between the GPUs, a device this process cannot open ...)
                if dist is None:
                    raise
                c_ok = [False, f"{type(e).__name__}: {e}"]
        else:
            reset = solve = sync = (lambda: None)
        if dist is not None:
            # every rank has to agree on the engine: without peer access fall back to one process per GPU over RCCL
            dist.broadcast_object_list(c_ok, src=0)
        for b in mine:
            try:
                os.unlink(f"{tag}_{b}.npy")
            except OSError:
                pass
        parallelism = (f"row-tiled x{nband}: C engine (j2p_tiled), one process drives all GPUs, one host thread per band; "
                       "edge rows and norm row sums read over peer access, ordered by HIP events")
    fallback_note = ""
    if c_tiled and not c_ok[0]:
        c_tiled = False
        fallback_note = f"; C engine unavailable ({c_ok[1]})"
        if rank == 0:
            print(f"bench: C row tiling unavailable, using the RCCL harness: {c_ok[1]}", file=sys.stderr, flush=True)
    if tiled_mode and not c_tiled:
        from jpeg2png_amd import tiled
        if dist is None:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        W = a.size or 16384
        rows_per_gpu = 2048 if not a.size else max(64, a.size // 8 // 16 * 16)
        H = rows_per_gpu * n_gpus
        its = a.iterations or 100
        seed = 1234 + 4
        workload = (f"{W}x{H} Y-only Q10 -i {its}, row-tiled {rows_per_gpu} rows/GPU over {n_gpus} GPUs "
                    f"(BASELINE configs[3] at 8 GPUs)")
        r0, r1 = rank * rows_per_gpu, (rank + 1) * rows_per_gpu
        band_planes = synth.make_planes(W, H, "444", 10, seed=seed, y_only=True, rows=(r0, r1))
        for p in band_planes:
            p.h = H                       # planes describe the whole image; arrays are band-local
        engine = tiled.HipBandEngine(band_planes, WEIGHT, [PWEIGHT], its, (r0, r1), local_rank)
        del band_planes
        driver = tiled.RowTiledSolver(engine)
        _flush_c_stdio()

        def reset():
            engine.reset()
            driver.start()
        solve, sync = (lambda: driver.iterate(its)), engine.solver.sync
        eng = engine.solver
        reset()
        parallelism = (f"row-tiled x{n_gpus}, one process per GPU, RCCL halo send/recv + norm all-gather "
                       f"({'librccl called on the solver streams' if driver.direct is not None else 'through torch.distributed'})"
                       + fallback_note)

    def barrier():
        if dist is not None and dist.is_initialized():
            dist.barrier()

    for _ in range(a.warmup):
        reset()
        solve()
    sync()
    if eng is not None:
        eng.enable_timing(a.timing_every)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        reset()
        solve()
    sync()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    g_ms, p_ms, samples = eng.kernel_times() if eng is not None else (0.0, 0.0, 0)

    if dist is not None and dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_device else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        px = W * H
        value = px * its * a.steps / elapsed / 1e6
        ngp = n_gpus if n_gpus > 1 else 1
        band_px = W * rows_per_gpu
        # pixels per TIMED launch: in the row-tiled schedules the events bracket the interior launches only
        # (all 16-row segments but the band's first and last; all block rows but the first and last)
        px_of = {"k_gradient": band_px, "k_project": band_px}
        if tiled_mode and (c_tiled or driver.overlap):
            px_of = {"k_gradient": W * (rows_per_gpu - 32), "k_project": W * (rows_per_gpu - 16)}
        per_kernel = {}
        for kname, kms, kb in (("k_gradient", g_ms, BYTES_GRADIENT), ("k_project", p_ms, BYTES_PROJECT)):
            gbs = px_of[kname] * kb / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
            per_kernel[kname] = {"avg_launch_ms": round(kms, 4), "algorithmic_bytes_per_launch": px_of[kname] * kb,
                                 "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
        kern = min(per_kernel, key=lambda k: per_kernel[k]["frac"])
        # whole iteration, wall clock, per GPU (bands on one GPU share it)
        gpus_used = ngp
        it_gbs = BYTES_ITERATION * value * 1e6 / gpus_used / 1e9
        it_ms = elapsed / a.steps / its * 1e3
        # HBM bytes per launch from rocprofv3 PMC passes of this same workload (profiles/, corrected as
        # MI355X_MICROARCH.md prescribes); only meaningful for the N=1 workload they were taken on
        traffic, traffic_src = None, None
        for tag in ("r02", "r01"):
            pmc = os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json")
            if traffic is None and not tiled_mode and not a.size and os.path.exists(pmc):
                with open(pmc) as f:
                    summ = json.load(f)
                tot = {}
                for name, v in summ.items():
                    if isinstance(v, dict) and "hbm_bytes_per_launch" in v:
                        for kk in ("k_gradient", "k_project"):
                            if name.startswith("j2p::" + kk) or name.startswith(kk):
                                tot[kk] = v["hbm_bytes_per_launch"]
                if len(tot) == 2:
                    traffic = tot["k_gradient"] + tot["k_project"]
                    traffic_src = f"profiles/{tag}_pmc_summary.json (FETCH_SIZE x2 + WRITE_SIZE; k_gradient + k_project)"
        out = {
            "metric": "Mpixel-iterations/sec on 4K Y-plane", "value": round(value, 1),
            "unit": "Mpixel-iterations/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "iterations_per_step": its, "weight": WEIGHT, "pweight": PWEIGHT,
                       "parallelism": parallelism},
            "roofline": {"bound": "hbm", "scope": "whole iteration (k_gradient + k_project, launch gaps included), wall clock, per GPU",
                         "kernel": kern, "achieved": round(it_gbs, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(it_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_unit": "bytes per iteration", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_iteration": band_px * BYTES_ITERATION,
                         "iteration_ms": round(it_ms, 5),
                         "kernel_frac": per_kernel[kern]["frac"],
                         "per_kernel": per_kernel,
                         "event_samples": samples,
                         "note": "per-kernel durations come from HIP events around every "
                                 f"{a.timing_every}th iteration; the event records themselves cost time on those "
                                 "iterations, so the two durations can add up to more than iteration_ms"},
        }
        if not tiled_mode and not a.no_other_configs and not a.size:
            eng.close()
            out["other_configs"] = other_configs(j, synth)
        if not tiled_mode and not a.no_cpu_baseline:
            out["cpu_baseline"], out["cpu_baseline_all_cores"] = cpu_baseline(W, seed)
        _flush_c_stdio()
        print(json.dumps(out), flush=True)
    if dist is not None and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
