"""GPU parity: HIP path (through the C-ABI) vs the CPU oracle on identical inputs.

Bar (BASELINE.json north_star): per-plane PSNR >= 80 dB, peak 255.  The kernels
reproduce the reference arithmetic operation for operation, so the expectation
checked here is stronger: BIT-IDENTICAL planes.  The only tolerated source of a
difference is the summation order of the double-precision sum(g*g) (SURVEY.md
§7 hard part 2), which can flip the float norm by one ulp with probability
~1e-6 per iteration at these sizes; a case that is not bit-identical must still
clear 80 dB and is reported.
"""
import copy
import os

import numpy as np
import pytest

from conftest import band_devices, bit_equal, make_case, psnr

pytestmark = pytest.mark.gpu

PSNR_BAR_DB = 80.0   # stated tolerance for the floating-point path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dct_blocks_bit_exact(lib, oracle):
    import jpeg2png_amd as j
    rng = np.random.default_rng(7)
    b = np.concatenate([rng.normal(0, 60, (4096, 64)), rng.uniform(-1e-3, 1e-3, (64, 64)),
                        np.zeros((8, 64)), rng.integers(-1024, 1024, (512, 64))]).astype(np.float32)
    assert bit_equal(j.dct8x8_blocks(b, inverse=False), oracle.dct_blocks(b, inverse=False))
    assert bit_equal(j.dct8x8_blocks(b, inverse=True), oracle.dct_blocks(b, inverse=True))


def test_fast_division_and_sqrt_are_ieee(lib):
    """the gradient kernel's shared-reciprocal division and its square root must agree bit for
    bit with `/` and sqrtf() over the operand range the kernel screens for"""
    import jpeg2png_amd as j
    for seed in (1, 2, 3):
        assert j.math_selftest(1 << 26, seed=seed) == (0, 0)


def test_square_roots_exhaustively(lib):
    """both packed sqrt sequences equal sqrtf() on every float of the screened range"""
    import jpeg2png_amd as j
    assert j.sqrt_exhaustive() == (0, 0)


def test_short_division_exhaustively(lib):
    """phase B's one-correction division (div_exact_recip with an IEEE reciprocal) equals `/` for EVERY denominator
    mantissa against EVERY numerator mantissa — 2^46 quotients, about 16 s of GPU; scaling either operand by a power
    of two scales every step exactly and the kernels' operand screens keep all of it normal, so this is all cases"""
    import jpeg2png_amd as j
    bad, offenders = 0, []
    for first in range(0, 1 << 23, 1 << 16):
        b, off = j.division_exhaustive(3, first, 1 << 16)
        bad += b
        offenders += off
    assert bad == 0, offenders[:8]


def test_phase_a_short_division_fails_only_where_the_kernel_does_not_use_it(lib):
    """phase A refines the reciprocal of a norm twice from the v_rsq_f32 seed and corrects each quotient once.  Pass 1:
    that reciprocal is the correctly rounded 1 / n for EVERY norm except those with an all-ones mantissa — two radicands
    per binade pair, whose own mantissa ends in 0x7ffffe / 0x7fffff.  Pass 2 (sampled here; tools/division_exhaustive.py
    runs all 2^47 quotients, profiles/r03_division_exhaustive.jsonl): the one-correction quotient is `/` for every
    numerator except at those radicands.  k_gradient sends every row holding a radicand whose low 16 bits are >= 0xfffe
    down the IEEE path (allones_candidate), which covers them."""
    import jpeg2png_amd as j
    bad, offenders = j.division_exhaustive(1)
    assert bad == 226                                   # 2 radicands in each of the 113 binade pairs of [2^-100, 2^127)
    for bits in offenders:
        assert int(bits, 16) & 0xffff >= 0xfffe
    total = 1 << 24
    size = 1 << 14
    rng = np.random.default_rng(3)
    firsts = [0, (1 << 23) - size, 1 << 23, total - size] + [int(v) for v in rng.integers(0, total // size, 28) * size]
    for first in firsts:
        b, off = j.division_exhaustive(2, first, size)
        for o in off:
            assert (int(o, 16) >> 32) & 0xffff >= 0xfffe, f"slice {first}: {o}"
        if first != total - size:
            assert b == 0, f"slice {first}: {off}"
        else:
            assert 0 < b <= 8                           # the two radicands just below 4


def test_decode_plane_bit_exact(lib, oracle):
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    for (W, H) in ((64, 48), (200, 120), (8, 8)):
        p = synth.make_planes(W, H, "444", 10, seed=3, y_only=True)[0]
        assert bit_equal(j.decode_plane(p), oracle.decode_plane(p))


CASES = [
    # W, H, subsampling, quality, y_only, weight, pweight, iterations
    ("y_64x48", 64, 48, "444", 10, True, 0.3, 0.001, 12),
    ("y_tvonly", 72, 40, "444", 10, True, 0.0, 0.001, 10),
    ("y_noprob", 72, 40, "444", 10, True, 0.3, 0.0, 10),
    ("y_8x8", 8, 8, "444", 10, True, 0.3, 0.001, 5),
    ("y_ragged", 200, 136, "444", 25, True, 0.3, 0.001, 8),
    ("rgb444", 96, 64, "444", 10, False, 0.3, 0.001, 10),
    ("rgb420", 128, 96, "420", 10, False, 0.3, 0.001, 10),
    ("rgb420_padded", 40, 20, "420", 10, False, 0.3, 0.001, 10),
    ("rgb422", 80, 48, "422", 50, False, 0.3, 0.001, 6),
    ("rgb440", 48, 80, "440", 50, False, 0.3, 0.001, 6),
    ("rgb411", 96, 40, "411", 50, False, 0.3, 0.001, 6),
    ("rgb410_wide", 544, 48, "410", 25, False, 0.3, 0.001, 5),
    ("rgb420_wide", 400, 72, "420", 10, False, 0.3, 0.001, 8),
    ("rgb422_wide", 328, 40, "422", 10, False, 0.3, 0.001, 6),
    ("rgb440_wide", 200, 136, "440", 10, False, 0.3, 0.001, 6),
    ("y_512_50", 512, 512, "444", 10, True, 0.3, 0.001, 50),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_compute_matches_oracle(lib, oracle, case):
    import jpeg2png_amd as j
    name, W, H, sub, q, y_only, weight, pw, its = case
    planes = make_case(W, H, sub, q, seed=1234 + len(name), y_only=y_only)
    pws = [pw] * len(planes)
    want, want_log = oracle.oracle_compute(planes, weight, pws, its, log=True)
    got_planes = copy.deepcopy(planes)
    got_log = j.compute(got_planes, weight, pws, its, log=True)
    cw, ch = oracle.canvas_size(planes)
    exact = True
    for c, p in enumerate(got_planes):
        assert (p.w, p.h) == (cw, ch)                  # compute.c:459-460
        assert p.fdata.shape == want[c].shape
        db = psnr(p.fdata, want[c])
        assert db >= PSNR_BAR_DB, f"{name} channel {c}: PSNR {db:.1f} dB"
        exact &= bit_equal(p.fdata, want[c])
    assert exact, f"{name}: planes within {PSNR_BAR_DB} dB but not bit-identical"
    # log trace: tv / tv2 / prob_dist sums differ only by double summation order
    np.testing.assert_allclose(got_log, want_log, rtol=1e-9, atol=1e-9)
    assert got_log[0, 1] == 0.0                        # iteration 0: cos = d*q exactly (SURVEY §8c i)


def test_zero_iterations_returns_upsampled_input(lib, oracle):
    import jpeg2png_amd as j
    planes = make_case(64, 48, "420", 10, seed=5)
    want, _ = oracle.oracle_compute(planes, 0.3, [0.001] * 3, 0)
    got = copy.deepcopy(planes)
    j.compute(got, 0.3, [0.001] * 3, 0)
    for c in range(3):
        assert bit_equal(got[c].fdata, want[c])


def test_log_off_equals_log_on(lib):
    import jpeg2png_amd as j
    planes = make_case(96, 64, "420", 10, seed=9)
    a, b = copy.deepcopy(planes), copy.deepcopy(planes)
    j.compute(a, 0.3, [0.001] * 3, 7, log=False)
    j.compute(b, 0.3, [0.001] * 3, 7, log=True)
    for c in range(3):
        assert bit_equal(a[c].fdata, b[c].fdata)


def test_reset_and_device_decode(lib, oracle):
    """fdata=NULL makes the library decode on the device; reset() restarts from resident inputs."""
    import jpeg2png_amd as j
    planes = make_case(128, 64, "444", 10, seed=11, y_only=True)
    want, _ = oracle.oracle_compute(planes, 0.3, [0.001], 6)
    nof = copy.deepcopy(planes)
    nof[0].fdata = None
    with j.Solver(nof, 0.3, [0.001], 6) as s:
        s.run(6)
        first = s.download(0)
        s.reset()
        s.run(6)
        second = s.download(0)
    assert bit_equal(first, want[0])
    assert bit_equal(second, want[0])


@pytest.mark.parametrize("shape", [(264, 200, "420", 10), (520, 136, "444", 50), (136, 72, "411", 30)])
def test_narrow_coefficients_leave_the_bits_alone(lib, oracle, shape):
    """Channels whose quantised coefficients all fit [-127, 127] stay resident as one byte each and k_project reads those
    (J2P_OPT_NARROW_COEFFICIENTS): the same planes as with the int16 form and as the reference's; a channel with one
    larger coefficient keeps the int16 form, channel by channel; +-127 are still narrow, -128 is not."""
    import jpeg2png_amd as j
    W, H, sub, q = shape
    planes = make_case(W, H, sub, q, seed=21)
    pw = [0.001, 0.01, 0.0]
    assert all(np.abs(p.data).max() <= 127 for p in planes)
    planes[1].data[5] = 127
    planes[1].data[70] = -127
    for p in planes:
        p.fdata = oracle.decode_plane(p)
    want, _, _ = oracle.ref_compute(planes, 0.3, pw, 9)
    with j.Solver(planes, 0.3, pw, 9) as s:
        assert [s.coefficient_bytes(c) for c in range(3)] == [1, 1, 1]
        s.run(9)
        narrow = [s.download(c) for c in range(3)]
        s.reset()
        s.debug_option(j.J2P_OPT_NARROW_COEFFICIENTS, 0)
        assert [s.coefficient_bytes(c) for c in range(3)] == [2, 2, 2]
        s.run(9)
        wide = [s.download(c) for c in range(3)]
    for c in range(3):
        assert bit_equal(narrow[c], want[c]), f"channel {c}, one byte per coefficient"
        assert bit_equal(wide[c], want[c]), f"channel {c}, int16"
    # one coefficient out of range: that channel alone keeps the int16 form
    for value in (128, -128, 1000):
        big = copy.deepcopy(planes)
        big[2].data[64 * 3 + 9] = value
        for p in big:
            p.fdata = oracle.decode_plane(p)
        want, _, _ = oracle.ref_compute(big, 0.3, pw, 4)
        with j.Solver(big, 0.3, pw, 4) as s:
            assert [s.coefficient_bytes(c) for c in range(3)] == [1, 1, 2], value
            s.debug_option(j.J2P_OPT_NARROW_COEFFICIENTS, 1)
            assert [s.coefficient_bytes(c) for c in range(3)] == [1, 1, 2], value      # the option cannot narrow what does not fit
            s.run(4)
            for c in range(3):
                assert bit_equal(s.download(c), want[c]), (value, c)


def test_projection_property_full_size(lib, oracle):
    """size-independent property at a larger size: after every projection all DCT coefficients of
    the returned plane lie inside their quantisation interval (SURVEY.md §8c ii), up to the
    float rounding of one DCT/IDCT round trip."""
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    W = H = 1024
    p = synth.make_planes(W, H, "444", 10, seed=1237, y_only=True)[0]
    p.fdata = j.decode_plane(p)
    planes = [p]
    d = p.data.reshape(-1, 64).astype(np.float64)
    qt = p.quant_table.astype(np.float64)
    j.compute(planes, 0.3, [0.001], 20)
    out = planes[0].fdata
    assert np.isfinite(out).all()
    blocks = out.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    coefs = j.dct8x8_blocks(blocks, inverse=False).astype(np.float64)
    lo, hi = (d - 0.5) * qt, (d + 0.5) * qt
    slack = 1e-3
    assert (coefs >= lo - slack).all() and (coefs <= hi + slack).all()


def test_band_split_matches_whole(lib, oracle):
    """two row bands on one GPU, exchanging halos and norm partials through host copies, must
    reproduce the whole-canvas solver bit for bit (GPU-count invariant reduction order)."""
    import ctypes
    import jpeg2png_amd as j
    planes = make_case(128, 96, "444", 10, seed=21, y_only=True)
    its = 6
    with j.Solver(planes, 0.3, [0.001], its) as whole:
        whole.run(its)
        want = whole.download(0)
    hip = j.hip_runtime()
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    D2D = 3
    bands = [j.Solver(planes, 0.3, [0.001], its, band=(0, 48)), j.Solver(planes, 0.3, [0.001], its, band=(48, 96))]
    try:
        for _ in range(its):
            for s in bands:
                s.phase_gradient()
            infos = [s.exchange_info() for s in bands]
            for s in bands:
                s.sync()
            # all-gather of the per-tile-row partials
            for dst in infos:
                for src in infos:
                    n = src.local_tile_rows
                    hip.hipMemcpy(dst.partials_all + 8 * src.first_tile_row, src.partials_local, 8 * n, D2D)  # nch == 1
            hip.hipDeviceSynchronize()      # device-to-device hipMemcpy may return early; the solver streams are non-blocking
            for s in bands:
                s.phase_project()
            for s in bands:
                s.sync()
            infos = [s.exchange_info() for s in bands]
            nbytes = infos[0].halo_floats * 4
            hip.hipMemcpy(infos[1].recv_top[0], infos[0].send_bottom[0], nbytes, D2D)
            hip.hipMemcpy(infos[0].recv_bottom[0], infos[1].send_top[0], nbytes, D2D)
            hip.hipDeviceSynchronize()
        got = np.concatenate([s.download(0) for s in bands], axis=0)
    finally:
        for s in bands:
            s.close()
    assert bit_equal(got, want)


@pytest.mark.parametrize("mode", ["compute", "tiled:2", "tiled:3"])
def test_drop_in_compute_from_c(lib, oracle, tmp_path, mode):
    """the C `compute()` with the reference's signature, called from a C host program that
    provides logger_log / progressbar_inc like jpeg2png.c does: planes, canvas size rewrite,
    one log row and one progress tick per iteration (compute.c:428,272,449-452,455-461).
    mode tiled:<n>: the same call through j2p_compute_tiled with n row bands on device 0."""
    import os
    import struct
    import subprocess
    import jpeg2png_amd as j
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "dropin"
    subprocess.run(["gcc", "-O1", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c", "dropin_main.c"),
                    "-o", str(exe), j.LIB_PATH, "-Wl,-rpath," + os.path.dirname(j.LIB_PATH)], check=True)
    planes = make_case(72, 40 if mode == "compute" else 150, "420", 10, seed=31)
    its, weight, pw = 37, 0.3, [0.001, 0.001, 0.001]       # 37: not a multiple of the host chunk size
    blob = struct.pack("<IIf3f", len(planes), its, weight, *pw)
    for p in planes:
        blob += struct.pack("<4I", p.w, p.h, p.w_samp, p.h_samp)
        blob += np.ascontiguousarray(p.data, np.int16).tobytes() + np.ascontiguousarray(p.fdata, np.float32).tobytes()
        blob += np.ascontiguousarray(p.quant_table, np.uint16).tobytes()
    (tmp_path / "in.bin").write_bytes(blob)
    extra = [] if mode == "compute" else [mode]
    subprocess.run([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "log.csv"), *extra], check=True)
    want, want_log = oracle.oracle_compute(planes, weight, pw, its, log=True)
    raw = (tmp_path / "out.bin").read_bytes()
    ticks, tick_times = struct.unpack_from("<II", raw, 0)
    assert ticks == its                                     # progressbar_inc once per iteration (compute.c:449-452) ...
    assert tick_times >= min(its, 20)                       # ... and while it runs, not in a few bursts (compute_host.c: chunks follow the clock)
    off = 8
    cw, ch = oracle.canvas_size(planes)
    for c in range(len(planes)):
        w, h = struct.unpack_from("<II", raw, off)
        off += 8
        assert (w, h) == (cw, ch)
        got = np.frombuffer(raw, np.float32, w * h, off).reshape(h, w)
        off += 4 * w * h
        assert bit_equal(got, want[c])
    rows = np.loadtxt(tmp_path / "log.csv", delimiter=",", skiprows=1, usecols=(2, 3, 4, 5, 6), ndmin=2)
    assert rows.shape == (its, 5)
    assert np.array_equal(rows[:, 0], np.arange(its))
    np.testing.assert_allclose(rows[:, 1:], want_log, rtol=0, atol=1e-6 * max(1.0, np.abs(want_log).max()))


def test_drop_in_compute_dies_like_the_reference(lib, tmp_path):
    """precondition violations end in `jpeg2png: <message>` + exit(EXIT_FAILURE) (utils.c:20-28)"""
    import os
    import struct
    import subprocess
    import jpeg2png_amd as j
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "dropin"
    subprocess.run(["gcc", "-O1", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c", "dropin_main.c"),
                    "-o", str(exe), j.LIB_PATH, "-Wl,-rpath," + os.path.dirname(j.LIB_PATH)], check=True)
    blob = struct.pack("<IIf3f", 1, 3, 0.3, 0.001, 0, 0) + struct.pack("<4I", 8, 8, 1, 1)
    blob += np.zeros(64, np.int16).tobytes() + np.zeros(64, np.float32).tobytes() + np.zeros(64, np.uint16).tobytes()
    (tmp_path / "in.bin").write_bytes(blob)          # all-zero quant table: invalid (jpeg.c:41-45)
    r = subprocess.run([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "o"), str(tmp_path / "l")],
                       capture_output=True, text=True)
    assert r.returncode == 1
    assert r.stderr.startswith("jpeg2png: ")


def test_tiled_engine_on_one_gpu(exp_lib, oracle):
    """the row-tiling driver's HIP engine: device memory aliased as torch tensors, two bands on
    one GPU exchanging halos / partials through those tensors, and a world-size-1 RCCL group
    driving RowTiledSolver end to end.  Both must reproduce the whole-canvas solver bit for bit."""
    import os
    import torch
    import torch.distributed as dist
    import jpeg2png_amd as j
    from jpeg2png_amd import tiled
    W, H, its = 160, 128, 5
    planes = make_case(W, H, "444", 10, seed=41, y_only=True)
    with j.Solver(planes, 0.3, [0.001], its) as whole:
        want_rows = whole.run(its, log=True)
        want = whole.download(0)

    def band_planes(r0, r1):
        p = planes[0]
        d = p.data.reshape(p.h // 8, -1)[r0 // 8:r1 // 8].reshape(-1)
        return [j.Plane(p.w, p.h, 1, 1, d, p.quant_table, p.fdata[r0:r1])]

    # (a) two engines, exchanges done by hand through the aliased tensors
    bands = [(0, 64), (64, 128)]
    eng = [tiled.HipBandEngine(band_planes(*b), 0.3, [0.001], its, b, 0) for b in bands]
    try:
        def halo_swap():
            h0, h1 = eng[0].halo(), eng[1].halo()
            with eng[0].stream_context():
                torch.cuda.synchronize()
                h1["recv_top"][0].copy_(h0["send_bottom"][0])
                h0["recv_bottom"][0].copy_(h1["send_top"][0])
                torch.cuda.synchronize()
        halo_swap()
        for e in eng:
            e.commit_initial_halo()
        for _ in range(its):
            for e in eng:
                e.phase_gradient()
            torch.cuda.synchronize()
            allp = torch.cat([e.partials_local for e in eng])
            for e in eng:
                e.partials_all.copy_(allp)
            torch.cuda.synchronize()
            for e in eng:
                e.phase_project()
            halo_swap()
        got = np.concatenate([e.download(0) for e in eng], axis=0)
    finally:
        for e in eng:
            e.close()
    assert bit_equal(got, want)

    # (a') the same two bands with the gradient phase split (interior | edges on a side stream) and
    #      the halo copies on a third stream, exactly the event choreography of RowTiledSolver(overlap)
    eng = [tiled.HipBandEngine(band_planes(*b), 0.3, [0.001], its, b, 0) for b in bands]
    try:
        comm = torch.cuda.Stream()

        def halo_swap_async():
            done = [e.project_done_event() for e in eng]
            with torch.cuda.stream(comm):
                for d in done:
                    comm.wait_event(d)
                h0, h1 = eng[0].halo(), eng[1].halo()
                h1["recv_top"][0].copy_(h0["send_bottom"][0], non_blocking=True)
                h0["recv_bottom"][0].copy_(h1["send_top"][0], non_blocking=True)
                return comm.record_event()
        ready = halo_swap_async()
        for e in eng:
            e.stream.wait_event(ready)
            e.commit_initial_halo()
        ready = None
        for _ in range(its):
            for e in eng:
                assert e.can_split
                with e.stream_context():
                    e.gradient_interior()
                e.gradient_edges(ready)
            for e in eng:
                with e.stream_context():
                    e.finish_gradient()
            torch.cuda.synchronize()
            allp = torch.cat([e.partials_local for e in eng])
            for e in eng:
                with e.stream_context():
                    e.partials_all.copy_(allp)
                    e.phase_project()
            ready = halo_swap_async()
        torch.cuda.synchronize()
        got2 = np.concatenate([e.download(0) for e in eng], axis=0)
    finally:
        for e in eng:
            e.close()
    assert bit_equal(got2, want)

    # (a'') the schedule RowTiledSolver uses now, with two REAL neighbours: the projection's boundary block rows
    #       first, the halo rows copied between the two parts of the projection (exchange_info then has to point
    #       into the iterate being written), gradient interior, then the edge segments behind the copy
    eng = [tiled.HipBandEngine(band_planes(*b), 0.3, [0.001], its, b, 0) for b in bands]
    try:
        comm = torch.cuda.Stream()

        def halo_copy_after(events):
            with torch.cuda.stream(comm):
                for d in events:
                    comm.wait_event(d)
                h0, h1 = eng[0].halo(), eng[1].halo()
                h1["recv_top"][0].copy_(h0["send_bottom"][0], non_blocking=True)
                h0["recv_bottom"][0].copy_(h1["send_top"][0], non_blocking=True)
                return comm.record_event()
        ready = halo_copy_after([e.project_done_event() for e in eng])
        for e in eng:
            e.stream.wait_event(ready)
            e.commit_initial_halo()
        ready = None
        for _ in range(its):
            for e in eng:
                with e.stream_context():
                    e.gradient_interior()
                    e.gradient_edges_inline(ready)
                    e.finish_gradient()
            torch.cuda.synchronize()
            allp = torch.cat([e.partials_local for e in eng])
            for e in eng:
                with e.stream_context():
                    e.partials_all.copy_(allp)
                    e.project_boundary()
            ready = halo_copy_after([e.project_done_event() for e in eng])
            for e in eng:
                with e.stream_context():
                    e.project_interior()
        for e in eng:
            e.stream.wait_event(ready)
        torch.cuda.synchronize()
        got3 = np.concatenate([e.download(0) for e in eng], axis=0)
    finally:
        for e in eng:
            e.close()
    assert bit_equal(got3, want)

    # (b) the real driver over a 1-rank RCCL group
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29731")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        e = tiled.HipBandEngine(band_planes(0, H), 0.3, [0.001], its, (0, H), 0)
        drv = tiled.RowTiledSolver(e, log=True)       # logging on: the "+3 doubles" exchange of every iteration
        assert drv.overlap
        drv.start()
        drv.iterate(its)
        got1 = e.download(0)
        np.testing.assert_allclose(drv.log_rows(), want_rows, rtol=1e-12, atol=0)
        e.reset()
        drv2 = tiled.RowTiledSolver(e, overlap=False)
        drv2.start()
        drv2.iterate(its)
        got1b = e.download(0)
        # (c) the lone rank as its own neighbour: RCCL grouped send/recv on the aliased halo tensors and the
        #     real all-gather every iteration (the rows it receives land outside the image and are masked)
        os.environ["J2P_TILED_SELF_NEIGHBOURS"] = "1"
        try:
            got1c = []
            for overlap in (True, False):
                e.reset()
                drv3 = tiled.RowTiledSolver(e, overlap=overlap, log=True)
                assert drv3.self_neighbours and drv3.up == 0 and drv3.down == 0
                drv3.start()
                drv3.iterate(its)
                got1c.append(e.download(0))
                np.testing.assert_allclose(drv3.log_rows(), want_rows, rtol=1e-12, atol=0)
        finally:
            del os.environ["J2P_TILED_SELF_NEIGHBOURS"]
        e.close()
    finally:
        dist.destroy_process_group()
    assert bit_equal(got1, want)
    assert bit_equal(got1b, want)
    assert bit_equal(got1c[0], want) and bit_equal(got1c[1], want)


def test_out_of_range_operands_take_the_ieee_path(lib, oracle):
    """the kernels' short division / sqrt sequences are only used inside a screened operand range;
    inputs outside it (tiny and huge pixel values, a 16-bit quantisation table whose q*q exceeds
    2^26) must fall back to the plain IEEE forms and still match the oracle bit for bit"""
    import jpeg2png_amd as j
    planes = make_case(136, 72, "420", 10, seed=77)
    rng = np.random.default_rng(5)
    for p in planes:
        f = p.fdata
        idx = rng.integers(0, f.size, 40)
        f.reshape(-1)[idx[:10]] = 1e-30          # 0 < |y| < 2^-20: differences far below 2^-44
        f.reshape(-1)[idx[10:20]] = -3e-39       # subnormal
        f.reshape(-1)[idx[20:30]] = 7e-8
        f.reshape(-1)[idx[30:]] = 3e13           # > 2^41
    q = planes[1].quant_table.copy()
    q[5:9] = [9000, 20000, 30000, 12345]         # q*q > 2^26: phase B's table path must switch off
    # (q >= 32768 is avoided: there the reference's own SSE2 and C builds disagree)
    planes[1].quant_table = q
    for joint in (True, False):
        sel = planes if joint else planes[1:2]
        pws = [0.001] * len(sel)
        want, want_log = oracle.oracle_compute(sel, 0.3, pws, 6, log=True)
        got = copy.deepcopy(sel)
        got_log = j.compute(got, 0.3, pws, 6, log=True)
        for c in range(len(sel)):
            assert np.isfinite(want[c]).all()
            assert bit_equal(got[c].fdata, want[c]), f"joint={joint} channel {c}"
        np.testing.assert_allclose(got_log, want_log, rtol=1e-9, atol=1e-9)


def test_noise_around_zero_takes_the_ieee_path(exp_lib, oracle, monkeypatch):
    """all-zero coefficient blocks (a flat grey area: chroma 0) leave rounding noise of 1e-17 and
    below around 0 for the first iterations — far under the 2^-20 the short division / sqrt
    sequences are screened for, down to values whose squares are subnormal.  Such rows must take
    the IEEE path, in both schedules of the joint gradient kernel and in the 1-channel kernel."""
    import jpeg2png_amd as j
    from oracle import bindings
    planes = make_case(264, 136, "420", 10, seed=99)
    for c, p in enumerate(planes):
        d = p.data.reshape(p.h // 8, p.w // 8, 64)
        if c == 0:
            d[8:11, 14:20] = 0                   # luma flat too under part of the grey patch
        else:
            d[2:7, 3:13] = 0
        p.fdata = bindings.decode_plane(p)
    pws = [0.001] * 3
    for its in (3, 6):
        want, want_log = oracle.oracle_compute(planes, 0.3, pws, its, log=True)
        if its == 3:
            tiny = sum(int(((np.abs(w) < 2.0 ** -43) & (w != 0)).sum()) for w in want)
            assert tiny > 1000, "the case no longer produces noise around 0"
        for mode in ("0", "1"):
            monkeypatch.setenv("J2P_JOINT_INWAVE", mode)
            for log in (False, True):
                got = copy.deepcopy(planes)
                got_log = j.compute(got, 0.3, pws, its, log=log)
                for c in range(3):
                    assert bit_equal(got[c].fdata, want[c]), f"its {its} mode {mode} log {log} channel {c}"
                if log:
                    np.testing.assert_allclose(got_log, want_log, rtol=1e-9, atol=1e-9)
        monkeypatch.delenv("J2P_JOINT_INWAVE")
        for c in (0, 1):
            want1, _ = oracle.oracle_compute(planes[c:c + 1], 0.3, [0.001], its)
            got = copy.deepcopy(planes[c:c + 1])
            j.compute(got, 0.3, [0.001], its)
            assert bit_equal(got[0].fdata, want1[0]), f"its {its} separate channel {c}"


def test_both_joint_modes_agree(exp_lib, oracle, monkeypatch):
    """channels-in-one-wavefront and one-wavefront-per-channel gradient kernels are two schedules of
    the same arithmetic"""
    import jpeg2png_amd as j
    planes = make_case(264, 88, "420", 10, seed=88)
    want, _ = oracle.oracle_compute(planes, 0.3, [0.001] * 3, 7)
    for mode in ("0", "1"):
        monkeypatch.setenv("J2P_JOINT_INWAVE", mode)
        got = copy.deepcopy(planes)
        j.compute(got, 0.3, [0.001] * 3, 7)
        for c in range(3):
            assert bit_equal(got[c].fdata, want[c]), f"mode {mode} channel {c}"


@pytest.mark.parametrize("shape", [(520, 136, "420", False), (1000, 96, "444", True), (264, 200, "422", False)])
def test_every_schedule_switch_leaves_the_bits_alone(exp_lib, oracle, shape, monkeypatch):
    """the J2P_OPT_* switches select schedules of the same arithmetic (where the norm is reduced, whether g is
    streamed non-temporally, one projection launch or one per sampling class), J2P_PX / J2P_RPW the geometry of the
    gradient strips (one or two columns per lane, rows per strip): every combination the solver can pick by itself —
    the choice depends on the canvas size — must give the bits of the reference on ONE canvas"""
    import jpeg2png_amd as j
    w, h, sub, yonly = shape
    planes = make_case(w, h, sub, 10, seed=91, y_only=yonly)
    n = len(planes)
    its = 9
    want, _ = oracle.oracle_compute(planes, 0.3, [0.001] * n, its)
    settings = [
        {},                                                                         # the solver's own choice
        {j.J2P_OPT_NORM_IN_PROJECT: 0, j.J2P_OPT_NORM_FOLD: 0},                     # stand-alone norm kernel
        {j.J2P_OPT_NORM_IN_PROJECT: 0, j.J2P_OPT_NORM_FOLD: 1},                     # both levels inside k_gradient
        {j.J2P_OPT_NORM_IN_PROJECT: 0, j.J2P_OPT_NORM_FOLD: 1, j.J2P_OPT_NT_GRADIENT: 2, j.J2P_OPT_MIXED_PROJECT: 0},
        {j.J2P_OPT_NORM_FOLD: 1, j.J2P_OPT_NORM_IN_PROJECT: 1},                     # level 2 inside k_project, every wavefront
        {j.J2P_OPT_NORM_FOLD: 1, j.J2P_OPT_NORM_IN_PROJECT: 2, j.J2P_OPT_MIXED_PROJECT: 0},   # ... the workgroup's first wavefront
        {j.J2P_OPT_NORM_FOLD: 1, j.J2P_OPT_NORM_IN_PROJECT: 2, j.J2P_OPT_MIXED_PROJECT: 0, j.J2P_OPT_NT_GRADIENT: 3},
        {j.J2P_OPT_NT_GRADIENT: 1},                                                 # what working sets beyond the Infinity Cache get:
        {j.J2P_OPT_NT_GRADIENT: 2},                                                 # g / + prob state / + coefficients non-temporal
        {j.J2P_OPT_NT_GRADIENT: 3},
        {j.J2P_OPT_NT_GRADIENT: 3, j.J2P_OPT_NORM_IN_PROJECT: 0, j.J2P_OPT_NORM_FOLD: 0},
        {j.J2P_OPT_MIXED_PROJECT: 0},                                               # what a > 1 Mpixel canvas gets
        {j.J2P_OPT_MIXED_PROJECT: 0, j.J2P_OPT_NORM_IN_PROJECT: 0, j.J2P_OPT_NORM_FOLD: 0, "J2P_JOINT_INWAVE": "1"},
    ]
    for px in ("2", "1"):
        monkeypatch.setenv("J2P_PX", px)
        for opts in settings:
            if "J2P_JOINT_INWAVE" in opts:
                if px == "1":
                    continue                    # the in-wavefront joint kernel exists with two columns per lane only
                monkeypatch.setenv("J2P_JOINT_INWAVE", "1")
            with j.Solver(planes, 0.3, [0.001] * n, its) as s:
                for k, v in opts.items():
                    if not isinstance(k, str):
                        s.debug_option(k, v)
                s.run(its)
                for c in range(n):
                    assert bit_equal(s.download(c), want[c]), f"px {px} options {opts} channel {c}"
            monkeypatch.delenv("J2P_JOINT_INWAVE", raising=False)
    # rows per strip (what canvases of other sizes get) with both strip widths, whole canvas and bands
    for px in ("2", "1"):
        for rpw in ("4", "8", "16"):
            monkeypatch.setenv("J2P_PX", px)
            monkeypatch.setenv("J2P_RPW", rpw)
            got = copy.deepcopy(planes)
            rows = j.compute(got, 0.3, [0.001] * n, its, log=True)
            for c in range(n):
                assert bit_equal(got[c].fdata, want[c]), f"px {px} rpw {rpw} channel {c}"
            assert np.isfinite(rows).all()
            if h >= 128:
                with j.TiledSolver(planes, 0.3, [0.001] * n, its, devices=band_devices(2)) as t:
                    t.run(its)
                    for c in range(n):
                        assert bit_equal(t.download(c), want[c]), f"px {px} rpw {rpw}, two bands: channel {c}"
    # strip heights that do not divide the band alignment (what tools/rpw_fine.py sweeps): whole canvases take them,
    # band solvers keep their own choice — the bits are the reference's either way
    monkeypatch.setenv("J2P_PX", "2")
    for rpw in ("2", "6", "12", "24"):
        monkeypatch.setenv("J2P_RPW", rpw)
        got = copy.deepcopy(planes)
        j.compute(got, 0.3, [0.001] * n, its)
        for c in range(n):
            assert bit_equal(got[c].fdata, want[c]), f"rpw {rpw} channel {c}"
        if h >= 128:
            with j.TiledSolver(planes, 0.3, [0.001] * n, its, devices=band_devices(2)) as t:
                t.run(its)
                for c in range(n):
                    assert bit_equal(t.download(c), want[c]), f"rpw {rpw} (ignored by band solvers), two bands: channel {c}"


@pytest.mark.parametrize("shape", [(700, 328, 0.3), (1000, 96, 0.3), (264, 200, 0.0), (4096, 48, 0.3)])
def test_half_and_quarter_items_leave_the_bits_alone(exp_lib, oracle, shape, monkeypatch):
    """the first workgroups of a gradient launch march two tile rows at once, the last ones half and quarter tile rows
    (grad_item / march_rows in j2p_kernels.hip.h; the solver's own choice by launch size): who marches a row must not
    change a bit — every share of doubles, halves and quarters (in 1/256 of an XCD's run), on canvases whose last tile row is short (328 = 20 x 16 + 8),
    whole and as bands, with the CSV sums, top-down and bottom-up (what canvases past the Infinity Cache get), against
    the reference's bits"""
    import jpeg2png_amd as j
    w, h, weight = shape
    planes = make_case(w, h, "444", 10, seed=97, y_only=True)
    its = 9
    want, want_rows = oracle.oracle_compute(planes, weight, [0.001], its, log=True)
    monkeypatch.setenv("J2P_PX", "2")
    monkeypatch.setenv("J2P_RPW", "16")
    for zd, zb, zc, rev in ((0, 0, 0, 0), (0, 256, 0, 0), (0, 0, 256, 1), (0, 100, 100, 0), (0, 64, 32, 1), (0, 26, 10, 0), (0, 1, 255, 0), (0, 0, 0, 1),
                            (256, 0, 0, 0), (256, 0, 0, 1), (128, 64, 32, 0), (100, 0, 100, 1), (160, 32, 10, 0)):
        monkeypatch.setenv("J2P_ZONE_D", str(zd))                # shares of double / half / quarter tile-row items
        monkeypatch.setenv("J2P_ZONE_B", str(zb))
        monkeypatch.setenv("J2P_ZONE_C", str(zc))
        monkeypatch.setenv("J2P_GRAD_REVERSE", str(rev))         # the launch walks the canvas bottom-up (Geo::reverse)
        got = copy.deepcopy(planes)
        rows = j.compute(got, weight, [0.001], its, log=True)
        assert bit_equal(got[0].fdata, want[0]), f"zones {zd}/{zb}/{zc} reverse {rev}"
        assert np.allclose(rows, want_rows, rtol=1e-9, atol=1e-9), f"zones {zd}/{zb}/{zc}: CSV rows"
        got = copy.deepcopy(planes)
        j.compute(got, weight, [0.001], its)
        assert bit_equal(got[0].fdata, want[0]), f"zones {zd}/{zb}/{zc}, no logging"
        if h >= 96:
            with j.TiledSolver(planes, weight, [0.001], its, devices=band_devices(3 if h >= 200 else 2)) as t:
                t.run(its)
                assert bit_equal(t.download(0), want[0]), f"zones {zd}/{zb}/{zc}, bands"


def test_concurrent_calls_are_independent(lib, oracle):
    """compute() is entered concurrently by the reference's `-s` and multi-file OpenMP regions
    (jpeg2png.c:147,330): six simultaneous calls from six host threads must each return the
    oracle's planes"""
    import threading
    import jpeg2png_amd as j
    cases = [make_case(96 + 8 * i, 64, "420" if i % 2 else "444", 10, seed=50 + i) for i in range(6)]
    wants = [oracle.oracle_compute(p, 0.3, [0.001] * 3, 9)[0] for p in cases]
    gots = [copy.deepcopy(p) for p in cases]
    errs = []

    def work(i):
        try:
            j.compute(gots[i], 0.3, [0.001] * 3, 9)
        except Exception as e:     # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs
    for i in range(6):
        for c in range(3):
            assert bit_equal(gots[i][c].fdata, wants[i][c]), f"call {i} channel {c}"


@pytest.mark.parametrize("case", [("y_2048_100", 2048, 2048, "444", True, 100), ("rgb420_1080p_50", 1920, 1080, "420", False, 50)],
                         ids=lambda c: c[0])
def test_full_size_against_the_compiled_reference(lib, oracle, case):
    """at realistic sizes the checker is the UNMODIFIED reference itself (oracle/_ref, a few seconds
    of CPU): 2048x2048 Y for 100 iterations and a 1080p 4:2:0 image (padded 1920x1088 canvas, joint)
    for 50.  Planes must clear the 80 dB bar; bit-identity is expected and reported."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    name, W, H, sub, y_only, its = case
    planes = synth.make_planes(W, H, sub, 10, seed=4321, y_only=y_only)
    for p in planes:
        p.fdata = j.decode_plane(p)
    pws = [0.001] * len(planes)
    want, want_log, _ = oracle.ref_compute(planes, 0.3, pws, its, log=True)
    got = copy.deepcopy(planes)
    got_log = j.compute(got, 0.3, pws, its, log=True)
    exact = True
    for c in range(len(planes)):
        db = psnr(got[c].fdata, want[c])
        assert db >= PSNR_BAR_DB, f"{name} channel {c}: PSNR {db:.1f} dB"
        exact &= bit_equal(got[c].fdata, want[c])
    assert exact, f"{name}: above {PSNR_BAR_DB} dB but not bit-identical (norm rounding flip?)"
    # tv, tv2 (and prob_dist) against the reference's CSV (6 decimals)
    np.testing.assert_allclose(got_log[:, 1:], want_log[:, 1:], rtol=1e-9, atol=2e-6)


def _sweep_check(lib, oracle, cs):
    import jpeg2png_amd as j
    planes = cs.planes()
    for p in planes:
        p.fdata = j.decode_plane(p)
    want, want_log, _ = oracle.ref_compute(planes, cs.weight, cs.pweights, cs.iterations, log=cs.log)
    got = copy.deepcopy(planes)
    got_log = j.compute(got, cs.weight, cs.pweights, cs.iterations, log=cs.log)
    for c in range(len(planes)):
        assert bit_equal(got[c].fdata, want[c]), f"sweep case {cs.describe()} channel {c}"
    if cs.log and cs.iterations:
        np.testing.assert_allclose(got_log[:, 1:], want_log[:, 1:], rtol=1e-9, atol=2e-6)


def test_randomised_sweep_against_the_compiled_reference(lib, oracle):
    """40 random configurations (tests/sweep_cases.py: ragged sizes from 1x1 up, all samplings, qualities,
    TV-only, per-channel pweights, flat areas, logging) against the unmodified reference, bitwise;
    tools/sweep_vs_ref.py runs the same stream for as long as one likes (400 cases of seed 2 pass)."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    from sweep_cases import cases
    for cs in cases(7, 40):
        _sweep_check(lib, oracle, cs)


@pytest.mark.parametrize("index", [55, 101])
def test_sign_of_zero_of_a_resampled_full_resolution_plane(lib, oracle, index):
    """seed-2 sweep cases in which the luma plane is smaller than the canvas (chroma pads further), so the
    reference resamples it with a 1x1 footprint: x_new = (x - (0.f + x)) + p turns a projected -0 into +0
    (compute.c:348-370,390-403).  Invisible in PSNR, but the planes are compared bitwise."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    from sweep_cases import case
    _sweep_check(lib, oracle, case(2, index))


def test_randomised_band_splits_match_the_whole_canvas(lib):
    """tools/sweep_bands.py as a test: 25 configurations of the shared case stream, each cut into 2-4 row
    bands at random aligned positions on one GPU, equal the whole-canvas solve bitwise (141 of 141 in the
    150-case run of seed 3)."""
    import subprocess
    import sys
    for extra in ([], ["--split"]):        # whole projection phase / boundary block rows, halo copy, interior
        env = dict(os.environ)
        if extra:                           # the split phases live in the experiments build
            from jpeg2png_amd.buildlib import build_experiments
            env["J2P_LIBRARY"] = build_experiments()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_bands.py"), "25", "9", *extra], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        assert "band splits bit-identical" in r.stdout


@pytest.mark.parametrize("order", ["lib_first", "torch_first"])
def test_library_and_torch_share_one_hip_runtime(order):
    """PyTorch-ROCm bundles its own HIP runtime; a fresh process that touches our library first and torch
    afterwards (or the other way round) must see the GPU from both"""
    import subprocess
    import sys
    code = f"""
import sys
sys.path.insert(0, {ROOT!r})
import jpeg2png_amd as j
from jpeg2png_amd import synth
def use_lib():
    planes = synth.make_planes(64, 48, "444", 10, seed=1, y_only=True)
    with j.Solver(planes, 0.3, [0.001], 2) as s:
        s.run(2); s.sync()
def use_torch():
    import torch
    torch.cuda.init()
    assert torch.cuda.device_count() >= 1
    (torch.ones(4, device="cuda") * 2).sum().item()
for f in ([use_lib, use_torch] if {order!r} == "lib_first" else [use_torch, use_lib]):
    f()
print("both fine")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "both fine" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_extreme_shapes_against_the_compiled_reference(lib, oracle):
    """tools/extreme_shapes.py as a test: 65500 pixels in one dimension (the JPEG limit) by a few in the other,
    all samplings, and two joint images of ~9.5 Mpixel"""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "extreme_shapes.py")], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def test_wide_randomised_sweep_against_the_compiled_reference(lib, oracle):
    """30 cases of the second stream (tests/sweep_cases.py: cases_wide): continuous weights, every quality,
    quantisation tables scaled into the 16-bit range, sparse coefficient data, up to 120 iterations
    (tools/sweep_wide.py: 1000 cases of seeds 1-3 pass)"""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    import jpeg2png_amd as j
    from sweep_cases import cases_wide, planes_wide
    for cs in cases_wide(5, 30):
        planes = planes_wide(cs)
        for p in planes:
            p.fdata = j.decode_plane(p)
        want, _, _ = oracle.ref_compute(planes, cs.weight, cs.pweights, cs.iterations)
        got = copy.deepcopy(planes)
        j.compute(got, cs.weight, cs.pweights, cs.iterations)
        for c in range(len(planes)):
            assert bit_equal(got[c].fdata, want[c]), f"wide sweep case {cs.describe()} channel {c}"


@pytest.mark.parametrize("inwave", ["0", "1"])
def test_two_channel_joint_against_the_compiled_reference(lib, oracle, inwave):
    """nchannel == 2 (compute.c:118 allows 1..3; the CLI only uses 1 and 3): every pair of components of the
    sweep stream's colour cases, both schedules of the joint gradient kernel"""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    import subprocess
    import sys
    env = dict(os.environ, J2P_JOINT_INWAVE=inwave)
    if inwave == "1":               # that schedule lives in the experiments build
        from jpeg2png_amd.buildlib import build_experiments
        env["J2P_LIBRARY"] = build_experiments()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "two_channel.py")], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
