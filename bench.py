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
          (N = 8 is exactly the 16384x16384 config), one RCCL halo exchange + one
          all-gather of norm partials per iteration; weak scaling (fixed rows per GPU).

value = canvas pixels x iterations x steps / wall time over all ranks (max over ranks).
The JSON also carries
  roofline     : the slower of the two phase kernels, algorithmic bytes (SURVEY.md §8d:
                 gradient 16 B/px, step+projection 22 B/px, 38 B/px-iteration together)
                 over its average duration from HIP events recorded on the solver's
                 stream during the timed region, against the 8 TB/s HBM peak;
  cpu_baseline : the UNMODIFIED reference (oracle/_ref, built from /root/reference by
                 oracle/Makefile) — or our C port if that .so is absent — timed on this
                 box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_GRADIENT = 16              # per canvas pixel per launch (phase A)
BYTES_PROJECT = 22               # phase B
WEIGHT, PWEIGHT = 0.3, 0.001     # jpeg2png.c:22-23 defaults


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=0, help="override plane width (debug)")
    ap.add_argument("--height", type=int, default=0, help="override plane height (debug, single GPU)")
    ap.add_argument("--iterations", type=int, default=0, help="override iterations per solve (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timing-every", type=int, default=16, help="HIP-event sample stride (iterations)")
    ap.add_argument("--force-tiled", action="store_true", help="run the row-tiled path even with one rank (debug)")
    return ap.parse_args()


def _flush_c_stdio():
    """RCCL prints its NCCL_DEBUG=VERSION banner through C stdio, which is block-buffered on a pipe and would
    otherwise surface at exit, after the JSON line: push it out when the communicators exist"""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


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
    return {"value": round(width * rows * its / secs / 1e6, 2), "unit": "Mpixel-iterations/s", "cores": 1,
            "kind": kind,
            "sample": f"{width}x{rows} Y-only Q10, {its} iterations, weight {WEIGHT}, pweight {PWEIGHT}, "
                      f"{secs:.2f} s inside compute(), host has {os.cpu_count()} cores"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("J2P_BENCH_ONE_DEVICE"):      # debugging on a 1-GPU box: every rank on the same device
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

    tiled_mode = n_gpus > 1 or a.force_tiled
    if not tiled_mode:
        W = a.size or 4096
        H = a.height or W
        its = a.iterations or 500
        seed = 1234 + 3
        workload = f"{W}x{H} Y-only Q10 -i {its} (BASELINE configs[2])"
        planes = synth.make_planes(W, H, "444", 10, seed=seed, y_only=True)
        solver = j.Solver(planes, WEIGHT, [PWEIGHT], its, device=local_rank)   # fdata=None: decoded on device
        del planes

        def reset():
            solver.reset()

        def solve():
            solver.run(its)

        def sync():
            solver.sync()
        eng = solver
    else:
        import torch.distributed as dist
        from jpeg2png_amd import tiled
        if not dist.is_initialized():
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

        def solve():
            driver.iterate(its)

        def sync():
            engine.solver.sync()
        eng = engine.solver
        reset()

    def barrier():
        if tiled_mode:
            import torch.distributed as dist
            dist.barrier()

    for _ in range(a.warmup):
        reset()
        solve()
    sync()
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
    g_ms, p_ms, samples = eng.kernel_times()

    if tiled_mode:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        px = W * H
        value = px * its * a.steps / elapsed / 1e6
        band_px = px // n_gpus
        # the two phase kernels take the same time to within run-to-run noise; report the one with the
        # lower roofline fraction (k_gradient) unless k_project is clearly the longer one, and list both
        if g_ms >= 0.97 * p_ms:
            kern, dur_ms, bpp = "k_gradient", g_ms, BYTES_GRADIENT
        else:
            kern, dur_ms, bpp = "k_project", p_ms, BYTES_PROJECT
        achieved = band_px * bpp / (dur_ms * 1e-3) / 1e9 if dur_ms > 0 else 0.0
        # pixels per TIMED launch: in the row-tiled schedule the events bracket the interior launches only
        # (all 16-row segments but the band's first and last; all block rows but the first and last)
        px_of = {"k_gradient": band_px, "k_project": band_px}
        if tiled_mode and driver.overlap:
            px_of = {"k_gradient": W * (rows_per_gpu - 32), "k_project": W * (rows_per_gpu - 16)}
        achieved = px_of[kern] * bpp / (dur_ms * 1e-3) / 1e9 if dur_ms > 0 else 0.0
        per_kernel = {}
        for kname, kms, kb in (("k_gradient", g_ms, BYTES_GRADIENT), ("k_project", p_ms, BYTES_PROJECT)):
            gbs = px_of[kname] * kb / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
            per_kernel[kname] = {"avg_launch_ms": round(kms, 4), "algorithmic_bytes_per_launch": px_of[kname] * kb,
                                 "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
        # HBM bytes per launch from rocprofv3 PMC passes of this same workload (profiles/, corrected as
        # MI355X_MICROARCH.md prescribes); only meaningful for the N=1 workload they were taken on
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if not tiled_mode and not a.size and os.path.exists(pmc):
            with open(pmc) as f:
                summ = json.load(f)
            for name, v in summ.items():
                if isinstance(v, dict) and name.startswith("j2p::" + kern) and "hbm_bytes_per_launch" in v:
                    traffic, traffic_src = v["hbm_bytes_per_launch"], "profiles/r01_pmc_summary.json (FETCH_SIZE x2 + WRITE_SIZE)"
        out = {
            "metric": "Mpixel-iterations/sec on 4K Y-plane", "value": round(value, 1),
            "unit": "Mpixel-iterations/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "iterations_per_step": its, "weight": WEIGHT, "pweight": PWEIGHT,
                       "parallelism": "single GPU" if not tiled_mode else
                       f"row-tiled x{n_gpus}, RCCL halo send/recv + norm all-gather "
                       f"({'librccl called on the solver streams' if driver.direct is not None else 'through torch.distributed'})"},
            "roofline": {"bound": "hbm", "kernel": kern, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": px_of[kern] * bpp,
                         "avg_launch_ms": {"k_gradient": round(g_ms, 4), "k_project": round(p_ms, 4)},
                         "per_kernel": per_kernel,
                         "event_samples": samples,
                         "iteration_frac_38B": round(38.0 * value * 1e6 / n_gpus / 1e9 / HBM_PEAK_GBS, 4)},
        }
        if not tiled_mode and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(W, seed)
        _flush_c_stdio()
        print(json.dumps(out), flush=True)
    if tiled_mode:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
