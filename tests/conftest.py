import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


PARITY_NOTES = []


def parity_note(msg):
    """one line for the "parity" section pytest prints after the run (also with -q): what was compared with the
    reference and whether it was bit-identical — so that the driver's GPUTEST record says so"""
    PARITY_NOTES.append(msg)


# ---------------------------------------------------------------------------------------------------------------
# configs[3] at its stated parameters against the UNMODIFIED reference: 16384x16384 Y, `-i 100` is ~5 minutes of one
# CPU core inside the reference's compute() (its OpenMP regions gain nothing on one channel).  The
# GPU suite is about as long, so the reference run starts in a worker thread as soon as collection has finished —
# if the test that needs it was selected — and the test is moved to the end of the run, where it joins the thread.
# (ctypes releases the GIL during the call; the plane is synthesised by worker processes.)
# The run must not be able to turn a green suite red under `-x` on a slower or smaller lease: it is not started with
# less than 12 GiB of host memory available (it needs ~9), and the test waits for it only as long as the suite's time
# budget allows (J2P_GPU_SUITE_BUDGET_S, default 1100 s from the start of the session, less 120 s for the test's own GPU
# work).  Without the live run the test compares with the SAME plane all the same — through the digest of the
# reference's plane for this workload that tests/golden/bench_digests.json holds ("configs[3] N=8": generated from
# oracle/_ref by make_golden.py --bench-digests) — and says so, loudly, in the parity section.
# ---------------------------------------------------------------------------------------------------------------
_CONFIG3_TEST = "test_config3_full_size_i100_vs_reference_whole_and_8_bands"
_config3_job = {}


def _mem_available_gib():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / (1 << 20)
    except OSError:
        pass
    return None


def _config3_work():
    import time
    job = _config3_job
    try:
        import jpeg2png_amd as j
        from jpeg2png_amd import synth
        from oracle import bindings
        if not bindings.have_ref():
            job["skip"] = "oracle/_ref not built (needs /root/reference)"
            return
        # bench.py's configs[3] plane (same seed)
        plane = synth.make_y_plane_banded(16384, 16384, 10, seed=1234 + 4, band_rows=1024, workers=16)
        plane.fdata = j.decode_plane(plane)           # device decode: bit-exact vs jpeg.c:83-92 (test_decode_plane_bit_exact)
        job["plane"] = plane
        avail = _mem_available_gib()
        if avail is not None and avail < 12.0:
            job["no_reference"] = f"only {avail:.1f} GiB of host memory available (the reference's compute() on 16384x16384 needs ~9)"
            return
        t0 = time.perf_counter()
        want, _, secs = bindings.ref_compute([plane], 0.3, [0.001], 100)
        job["want"] = want
        job["seconds"] = secs if secs else time.perf_counter() - t0
    except BaseException as e:      # noqa: BLE001  (reported by the test that joins)
        job["error"] = e


def pytest_collection_finish(session):
    if _config3_job or not any(item.name == _CONFIG3_TEST for item in session.items):
        return
    if session.config.option.collectonly:
        return
    import threading
    import time
    th = threading.Thread(target=_config3_work, name="config3-reference", daemon=True)
    _config3_job["thread"] = th
    _config3_job["t_session"] = time.perf_counter()
    th.start()


def pytest_collection_modifyitems(config, items):
    last = [it for it in items if it.name == _CONFIG3_TEST]
    if last:
        items[:] = [it for it in items if it.name != _CONFIG3_TEST] + last


@pytest.fixture(scope="session")
def config3_reference():
    """callable -> (plane, reference planes or None, seconds inside the reference's compute(), seconds this call waited,
    why there are no reference planes or None)"""
    def join():
        import time
        job = _config3_job
        if "thread" not in job:                   # (the test was run in a way that skipped pytest_collection_finish)
            _config3_work()
        else:
            t0 = time.perf_counter()
            budget = float(os.environ.get("J2P_GPU_SUITE_BUDGET_S", "1100"))
            left = budget - (t0 - job["t_session"]) - 120.0
            job["thread"].join(timeout=max(left, 0.0))
            job["waited"] = time.perf_counter() - t0
            if job["thread"].is_alive():
                # (the plane itself is ready within the first minute; the thread goes on in the background and ends with the process)
                while "plane" not in job and "error" not in job and "skip" not in job:
                    time.sleep(0.5)
                if "want" not in job:
                    job["no_reference"] = (f"the reference's compute() had not finished {t0 - job['t_session'] + job['waited']:.0f} s into the "
                                           f"session (budget {budget:.0f} s, J2P_GPU_SUITE_BUDGET_S)")
        if "skip" in job:
            pytest.skip(job["skip"])
        if "error" in job:
            raise job["error"]
        return job["plane"], job.get("want"), job.get("seconds", 0.0), job.get("waited", 0.0), job.get("no_reference")
    return join


def pytest_terminal_summary(terminalreporter):
    if PARITY_NOTES:
        terminalreporter.section("parity vs the compiled reference")
        for line in PARITY_NOTES:
            terminalreporter.write_line(line)


@pytest.fixture(scope="session")
def lib():
    """The in-tree HIP library; built on demand (hipcc cross-compiles without a GPU)."""
    import jpeg2png_amd
    jpeg2png_amd.build()
    return jpeg2png_amd.load_library()


@pytest.fixture
def exp_lib(lib):
    """The experiments build of the library (-DJ2P_EXPERIMENTS: the schedules that lost their measurements and the
    environment knobs that select them) for the duration of one test: every Solver / TiledSolver / Batch created inside
    the test comes from it.  Built on demand (it travels with the tree when __graft_entry__.build() made it)."""
    import jpeg2png_amd
    from jpeg2png_amd.buildlib import build_experiments
    with jpeg2png_amd.library(build_experiments()) as l:
        yield l


@pytest.fixture(scope="session")
def oracle():
    from oracle import bindings
    bindings.oracle_lib()
    return bindings


def band_devices(n):
    """device ids for an n-band TiledSolver / an n-device Batch: the GPUs that are THERE, cycled to length n — on this
    pool's one-GPU boxes [0] * n (all bands share the GPU: everything of the multi-GPU path but the xGMI hop), on an 8-GPU
    node every exchange crosses real links and is compared with the reference / the whole-canvas solve just the same."""
    import jpeg2png_amd as j
    count = max(1, j.device_count())
    return [i % count for i in range(n)]


def psnr(a, b):
    """per-plane PSNR, peak 255."""
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return float("inf") if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def bit_equal(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32),
                                                 np.ascontiguousarray(b, np.float32).view(np.uint32))


def make_case(W, H, sub, quality, seed, y_only=False):
    """synthetic planes with fdata decoded by the CPU oracle (jpeg.c:83-92 + unbox)."""
    from jpeg2png_amd import synth
    from oracle import bindings
    planes = synth.make_planes(W, H, sub, quality, seed=seed, y_only=y_only)
    for p in planes:
        p.fdata = bindings.decode_plane(p)
    return planes
