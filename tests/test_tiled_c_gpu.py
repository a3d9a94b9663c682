"""The C-level row tiling (j2p_tiled / j2p_compute_tiled, jpeg2png_amd/csrc/j2p_tiled.hip) on ONE GPU: several
bands on device 0, one host thread each, ordered by events — everything of the multi-GPU path except the xGMI
hop itself.  The planes must equal the whole-canvas solver's bit for bit whatever the cut (the norm is a fixed
tree over the global array of 16-row tile sums), and the CSV rows must match to rounding."""
import copy

import numpy as np
import pytest

from conftest import bit_equal, make_case

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
    with j.TiledSolver(planes, 0.3, pws, its, devices=[0] * nband) as t:
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
        with j.TiledSolver(planes, weight, pws, its, devices=[0] * nb, cuts=cuts) as t:
            rows = np.concatenate([t.run(4, log=True), t.run(7, log=True)])
            got = [t.download(c) for c in range(len(planes))]
        for c in range(len(planes)):
            assert bit_equal(got[c], want[c]), f"trial {trial} cuts {cuts} channel {c}"
        np.testing.assert_allclose(rows, want_rows, rtol=1e-9, atol=1e-12)
        with j.TiledSolver(planes, weight, pws, its, devices=[0] * nb, cuts=cuts) as t:
            t.run(5)
            t.run(6)
            for c in range(len(planes)):
                assert bit_equal(t.download(c), want[c]), f"trial {trial} (no log) cuts {cuts} channel {c}"


def test_bad_cuts_are_rejected(lib):
    import jpeg2png_amd as j
    planes = make_case(64, 96, "420", 10, seed=3)
    with pytest.raises(j.J2PError, match="aligned|cuts"):
        j.TiledSolver(planes, 0.3, [0.001] * 3, 4, devices=[0, 0], cuts=[0, 40, 96])
    with pytest.raises(j.J2PError, match="bands"):
        j.TiledSolver(planes, 0.3, [0.001] * 3, 4, devices=[0] * 20)


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
