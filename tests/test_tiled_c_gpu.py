"""The C-level row tiling (j2p_tiled / j2p_compute_tiled, jpeg2png_amd/csrc/j2p_tiled.hip) on ONE GPU: several
bands on device 0, one host thread each, ordered by events — everything of the multi-GPU path except the xGMI
hop itself.  The planes must equal the whole-canvas solver's bit for bit whatever the cut (the norm is a fixed
tree over the global array of 16-row tile sums), and the CSV rows must match to rounding."""
import copy

import numpy as np
import pytest

from conftest import band_devices, bit_equal, make_case

pytestmark = pytest.mark.gpu


def whole_canvas(planes, weight, pws, its, log=False):
    import jpeg2png_amd as j
    ref = copy.deepcopy(planes)
    rows = j.compute(ref, weight, pws, its, log=log)
    return [p.fdata for p in ref], rows


@pytest.mark.parametrize("sub,y_only,nband", [("444", True, 2), ("444", True, 5), ("420", False, 3), ("422", False, 2),
                                              ("440", False, 4)])
def test_equal_bands_match_whole_canvas(lib, sub, y_only, nband):
    import jpeg2png_amd as j
    planes = make_case(200, 330, sub, 10, seed=77, y_only=y_only)
    pws = [0.001] * len(planes)
    its = 9
    want, want_rows = whole_canvas(planes, 0.3, pws, its, log=True)
    with j.TiledSolver(planes, 0.3, pws, its, devices=band_devices(nband)) as t:
        rows = t.run(its, log=True)
        for c in range(len(planes)):
            assert bit_equal(t.download(c), want[c]), f"channel {c}"
    np.testing.assert_allclose(rows, want_rows, rtol=1e-9, atol=1e-12)


def test_random_cuts_and_chunked_runs(lib):
    """bands of unequal height (down to a single 16-row segment: those use the unsplit phases), iterations issued
    in several run() calls, with and without logging"""
    import jpeg2png_amd as j
    from jpeg2png_amd import tiled
    rng = np.random.default_rng(5)
    for trial in range(8):
        sub = ["444", "420", "422", "440"][trial % 4]
        planes = make_case(int(rng.integers(40, 300)), int(rng.integers(100, 420)), sub, int(rng.integers(5, 60)),
                           seed=100 + trial, y_only=trial % 3 == 0)
        pws = [0.001] * len(planes)
        weight = 0.0 if trial == 5 else 0.3
        align = tiled.band_alignment(planes)
        H = max(p.h * p.h_samp for p in planes)
        units = (H + align - 1) // align
        nb = int(rng.integers(2, min(6, units) + 1))
        cuts = [0] + sorted((rng.choice(np.arange(1, units), nb - 1, replace=False) * align).tolist()) + [H]
        its = 11
        want, want_rows = whole_canvas(planes, weight, pws, its, log=True)
        with j.TiledSolver(planes, weight, pws, its, devices=band_devices(nb), cuts=cuts) as t:
            rows = np.concatenate([t.run(4, log=True), t.run(7, log=True)])
            got = [t.download(c) for c in range(len(planes))]
        for c in range(len(planes)):
            assert bit_equal(got[c], want[c]), f"trial {trial} cuts {cuts} channel {c}"
        np.testing.assert_allclose(rows, want_rows, rtol=1e-9, atol=1e-12)
        with j.TiledSolver(planes, weight, pws, its, devices=band_devices(nb), cuts=cuts) as t:
            t.run(5)
            t.run(6)
            for c in range(len(planes)):
                assert bit_equal(t.download(c), want[c]), f"trial {trial} (no log) cuts {cuts} channel {c}"


def test_bad_cuts_are_rejected(lib):
    import jpeg2png_amd as j
    planes = make_case(64, 96, "420", 10, seed=3)
    with pytest.raises(j.J2PError, match="aligned|cuts"):
        j.TiledSolver(planes, 0.3, [0.001] * 3, 4, devices=band_devices(2), cuts=[0, 40, 96])
    with pytest.raises(j.J2PError, match="bands"):
        j.TiledSolver(planes, 0.3, [0.001] * 3, 4, devices=band_devices(20))


def test_norm_fold_on_and_off_agree(lib):
    """the gradient-norm reduction inside k_gradient (last-arriving wavefronts) against the stand-alone reduction
    kernels of round 1: same bits, with one channel, with three, and over bands"""
    import jpeg2png_amd as j
    for sub, y_only in (("444", True), ("420", False)):
        planes = make_case(300, 200, sub, 10, seed=9, y_only=y_only)
        pws = [0.001] * len(planes)
        outs = []
        for fold in (1, 0):
            with j.Solver(planes, 0.3, pws, 15) as s:
                s.debug_option(j.J2P_OPT_NORM_FOLD, fold)
                s.run(15)
                outs.append([s.download(c) for c in range(len(planes))])
        for c in range(len(planes)):
            assert bit_equal(outs[0][c], outs[1][c])


def test_one_band_is_a_whole_canvas_solver(lib):
    """nband == 1 (the C API and TiledSolver(devices=[0]) reach it): a canvas above 2.5 Mpixel, whose whole-canvas
    solver keeps its own norm reduction (no per-tile-row sums for k_norm_bands to read) — planes and CSV rows must
    equal the plain solver's"""
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    planes = synth.make_planes(2048, 1536, "444", 10, seed=91, y_only=True)       # 3.1 Mpixel
    its = 6
    with j.Solver(planes, 0.3, [0.001], its) as s:
        want_rows = s.run(its, log=True)
        want = s.download(0)
    with j.TiledSolver(planes, 0.3, [0.001], its, devices=[0]) as t:
        assert t.bands() == [(0, 0, 1536)]
        rows = np.concatenate([t.run(2, log=True), t.run(4, log=True)])
        assert bit_equal(t.download(0), want)
        t.reset()
        t.run(its)
        assert bit_equal(t.download(0), want)
    np.testing.assert_allclose(rows, want_rows, rtol=1e-9, atol=1e-12)
    assert np.isfinite(rows).all()


def test_norm_from_bands_refuses_a_solver_without_row_sums(lib):
    """the underlying guard: a whole-canvas solver above 2.5 Mpixel leaves no level-1 sums, so the band reduction
    must fail loudly instead of reducing uninitialised memory"""
    import ctypes
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    planes = synth.make_planes(2048, 1536, "444", 10, seed=91, y_only=True)
    with j.Solver(planes, 0.3, [0.001], 2) as s:
        s.phase_gradient()
        e = s.exchange_info()
        rs = (ctypes.c_void_p * 1)(e.partials_local)
        first, count = (ctypes.c_uint * 1)(0), (ctypes.c_uint * 1)(e.global_tile_rows)
        lib.j2p_solver_norm_from_bands.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p]
        assert lib.j2p_solver_norm_from_bands(s._h, 1, rs, first, count, 0, None) == -4      # J2P_ESTATE
        s.phase_project()


def test_unlogged_then_logged_runs_report_nan_for_the_unknown_distance(lib):
    """the prob distance entering a logged run is only known if the iterations before it were logged: NaN in the
    first row otherwise, exactly as j2p_solver_run reports it; tv / tv2 are exact either way"""
    import jpeg2png_amd as j
    planes = make_case(200, 330, "420", 10, seed=77)
    pws = [0.001] * 3
    _, want_rows = whole_canvas(planes, 0.3, pws, 9, log=True)
    with j.TiledSolver(planes, 0.3, pws, 9, devices=band_devices(3)) as t:
        t.run(4)
        rows = t.run(5, log=True)
    assert np.isnan(rows[0, 0]) and np.isnan(rows[0, 1])
    np.testing.assert_allclose(rows[0, 2:], want_rows[4, 2:], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(rows[1:], want_rows[5:], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("exchange,norm", [("direct", "root"), ("copy", "root"), ("copy", "all")])
def test_every_schedule_of_the_tiling_gives_the_same_bits(exp_lib, exchange, norm, monkeypatch):
    """the exchanges riding on the phase kernels as peer writes (direct: two launches per band and iteration, the
    default) or round 3's copy kernel + one band reducing ||g|| for all / every band for itself (copy;
    J2P_TILED_NORM=all): the same tree over the same array and the same kernels on the same rows, so the same planes
    and the same CSV rows — joint 4:2:0 (subsampled and resampled projection paths push their edge rows too)"""
    import jpeg2png_amd as j
    planes = make_case(264, 410, "420", 10, seed=78)
    pws = [0.001] * 3
    want, want_rows = whole_canvas(planes, 0.3, pws, 10, log=True)
    monkeypatch.setenv("J2P_TILED_EXCHANGE", exchange)
    monkeypatch.setenv("J2P_TILED_NORM", norm)
    with j.TiledSolver(planes, 0.3, pws, 10, devices=band_devices(5)) as t:
        assert t.exchange() == exchange
        t.run(10)
        for c in range(3):
            assert bit_equal(t.download(c), want[c]), f"{exchange}/{norm}: channel {c}"
        t.reset()
        rows = t.run(10, log=True)
        for c in range(3):
            assert bit_equal(t.download(c), want[c]), f"{exchange}/{norm}, logged: channel {c}"
    np.testing.assert_allclose(rows, want_rows, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("sub,y_only,W,H", [("444", True, 200, 330), ("444", False, 136, 200), ("422", False, 152, 264),
                                            ("440", False, 152, 264), ("420", False, 77, 301)])
def test_direct_exchange_on_every_projection_path(lib, sub, y_only, W, H, monkeypatch):
    """the edge rows a band's projection stores into its neighbours' halo rows, on every store path of k_project: 1x1
    register path, 2x1 / 1x2 / 2x2 subsampled, ragged canvases (generic and uncovered pixels), iterations issued in
    two run() calls"""
    import jpeg2png_amd as j
    planes = make_case(W, H, sub, 12, seed=79, y_only=y_only)
    pws = [0.001] * len(planes)
    want, _ = whole_canvas(planes, 0.3, pws, 8)
    monkeypatch.setenv("J2P_TILED_EXCHANGE", "direct")
    with j.TiledSolver(planes, 0.3, pws, 8, devices=band_devices(3)) as t:
        assert t.exchange() == "direct"
        t.run(3)
        t.run(5)
        for c in range(len(planes)):
            assert bit_equal(t.download(c), want[c]), f"channel {c}"


@pytest.mark.parametrize("wait,nband", [("all", 5), ("root", 5), ("collector", 5), ("counter", 2), ("counter", 5)])
@pytest.mark.timeout(180)
def test_direct_exchange_wait_modes(lib, wait, nband, monkeypatch):
    """how a band's projection learns that every band's gradient launch has finished (J2P_TILED_WAIT): waiting for the
    N - 1 events itself, for one event of a root band that waited for them, for one event of a collecting stream, or —
    no events at all — for one VALUE in host memory that the gradient kernels count up themselves (hipStreamWaitValue64 /
    hipStreamWriteValue64): orderings of the same launches, so the same bits; five bands, logged and not, reset in
    between.  The value form is for bands on GPUs of their own (a waiting stream blocks its hardware queue, and a
    device's streams share four): it runs with two bands here, and with five on one GPU the engine takes the event form"""
    import jpeg2png_amd as j
    planes = make_case(200, 330, "420", 10, seed=82)
    pws = [0.001] * 3
    want, want_rows = whole_canvas(planes, 0.3, pws, 9, log=True)
    monkeypatch.setenv("J2P_TILED_EXCHANGE", "direct")
    monkeypatch.setenv("J2P_TILED_WAIT", wait)
    devices = band_devices(nband)
    crowded = max(devices.count(d) for d in devices) > 2       # the value form needs (nearly) a GPU per band
    with j.TiledSolver(planes, 0.3, pws, 9, devices=devices) as t:
        assert t.exchange() == ("direct" if wait == "all" or (wait == "counter" and crowded) else f"direct, wait {wait}")
        t.run(9)
        for c in range(3):
            assert bit_equal(t.download(c), want[c]), f"{wait}: channel {c}"
        t.reset()
        rows = np.concatenate([t.run(4, log=True), t.run(5, log=True)])
        for c in range(3):
            assert bit_equal(t.download(c), want[c]), f"{wait}, logged: channel {c}"
    np.testing.assert_allclose(rows, want_rows, rtol=1e-9, atol=1e-12)


def test_tall_narrow_canvas_falls_back_to_the_copy_exchange(lib):
    """a canvas with more than 1024 tile rows (here 4-row tile rows: few wavefronts, so short strips) does not fit the
    tree k_project runs for linked bands: the engine takes the copy exchange by itself, same bits"""
    import jpeg2png_amd as j
    planes = make_case(64, 4128, "444", 10, seed=81, y_only=True)
    want, _ = whole_canvas(planes, 0.3, [0.001], 5)
    with j.TiledSolver(planes, 0.3, [0.001], 5, devices=band_devices(3)) as t:
        assert t.exchange() == "copy"
        t.run(5)
        assert bit_equal(t.download(0), want[0])


def test_rccl_exchange_with_one_band_as_its_own_neighbour(exp_lib, monkeypatch):
    """the RCCL transport of the C engine on ONE GPU: one band (RCCL wants a GPU per rank), driven through the band
    machinery — ncclCommInitAll, ncclAllGather of its row sums, grouped ncclSend / ncclRecv of its edge rows to
    itself (they land above / below the image, where the kernels mask) — against the plain whole-canvas solver,
    bitwise; two bands on one GPU are refused with the reason"""
    import jpeg2png_amd as j
    planes = make_case(264, 410, "420", 10, seed=80)
    pws = [0.001] * 3
    want, want_rows = whole_canvas(planes, 0.3, pws, 9, log=True)
    monkeypatch.setenv("J2P_TILED_EXCHANGE", "rccl")
    monkeypatch.setenv("J2P_TILED_SELF_NEIGHBOURS", "1")
    try:
        t = j.TiledSolver(planes, 0.3, pws, 9, devices=[0])
    except j.J2PError as e:
        if "librccl" in str(e) or "RCCL" in str(e):
            pytest.skip(f"no usable librccl on this box: {e}")
        raise
    with t:
        assert t.exchange() == "rccl"
        rows = np.concatenate([t.run(4, log=True), t.run(5, log=True)])
        for c in range(3):
            assert bit_equal(t.download(c), want[c]), f"channel {c}"
        t.reset()
        t.run(9)
        for c in range(3):
            assert bit_equal(t.download(c), want[c]), f"after reset: channel {c}"
    np.testing.assert_allclose(rows, want_rows, rtol=1e-9, atol=1e-12)
    with pytest.raises(j.J2PError, match="one GPU per band"):
        j.TiledSolver(planes, 0.3, pws, 9, devices=[0, 0])
