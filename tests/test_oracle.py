"""CPU tests of the oracle (no GPU): the C restatement against (a) the golden fixtures that
were produced by the UNMODIFIED reference and (b) the compiled reference itself where
/root/reference is available.  Expectation: bit-identical planes."""
import glob
import os

import numpy as np
import pytest

from conftest import bit_equal, make_case
from jpeg2png_amd.synth import Plane

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(f for f in glob.glob(os.path.join(GOLDEN, "*.npz")) if not f.endswith("dct_blocks.npz"))


def load_case(path):
    z = np.load(path)
    planes = []
    c = 0
    while f"geom{c}" in z:
        w, h, ws, hs = (int(v) for v in z[f"geom{c}"])
        planes.append(Plane(w, h, ws, hs, z[f"data{c}"], z[f"quant{c}"], z[f"fdata{c}"]))
        c += 1
    outs = [z[f"out{i}"] for i in range(c)]
    return planes, float(z["weight"]), [float(v) for v in z["pweight"]], int(z["iterations"]), outs, z["log"]


def test_golden_fixtures_present():
    assert len(CASES) >= 7


def test_dct_against_golden(oracle):
    z = np.load(os.path.join(GOLDEN, "dct_blocks.npz"))
    assert bit_equal(oracle.dct_blocks(z["blocks"], inverse=False), z["fdct"])
    assert bit_equal(oracle.dct_blocks(z["blocks"], inverse=True), z["idct"])


def test_dct_is_orthonormal(oracle):
    """dct8x8s o idct8x8s ~ id and DC = sum/8 (SURVEY.md §8c v)."""
    rng = np.random.default_rng(3)
    b = rng.normal(0, 50, (64, 64)).astype(np.float32)
    f = oracle.dct_blocks(b, inverse=False)
    np.testing.assert_allclose(oracle.dct_blocks(f, inverse=True), b, atol=2e-4)
    np.testing.assert_allclose(f[:, 0], b.sum(axis=1) / 8, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_oracle_matches_reference_golden(oracle, path):
    planes, weight, pw, its, want, want_log = load_case(path)
    # the decoded planes stored with the fixture came from the reference's idct8x8s
    for p in planes:
        assert bit_equal(oracle.decode_plane(p), p.fdata)
    got, log = oracle.oracle_compute(planes, weight, pw, its, log=True)
    for c in range(len(planes)):
        assert bit_equal(got[c], want[c]), f"channel {c}"
    # the reference prints %f (6 decimals)
    np.testing.assert_allclose(log, want_log, rtol=0, atol=6e-7 * max(1.0, np.abs(want_log).max() * 1e-6))
    assert log[0, 1] == 0.0                     # iteration 0: cos = d*q exactly


LIVE = [
    ("y_88x56", 88, 56, "444", 10, True, 0.3, 0.001, 9),
    ("rgb420_72x40", 72, 40, "420", 10, False, 0.3, 0.001, 7),
    ("rgb440_40x72", 40, 72, "440", 50, False, 0.1, 0.01, 5),
]


@pytest.mark.parametrize("case", LIVE, ids=[c[0] for c in LIVE])
def test_oracle_matches_compiled_reference(oracle, case):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    name, W, H, sub, q, y_only, weight, pw, its = case
    planes = make_case(W, H, sub, q, seed=77 + its, y_only=y_only)
    pws = [pw] * len(planes)
    got, log = oracle.oracle_compute(planes, weight, pws, its, log=True)
    want, want_log, _ = oracle.ref_compute(planes, weight, pws, its, log=True)
    for c in range(len(planes)):
        assert bit_equal(got[c], want[c])
    np.testing.assert_allclose(log, want_log, rtol=0, atol=1e-6 * max(1.0, np.abs(want_log).max() * 1e-6))


def test_reference_dct_matches_oracle_dct(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(11)
    b = rng.normal(0, 80, (256, 64)).astype(np.float32)
    for inv in (False, True):
        assert bit_equal(oracle.dct_blocks(b, inv, "oracle"), oracle.dct_blocks(b, inv, "ref"))


def test_projection_keeps_coefficients_in_their_intervals(oracle):
    """after every projection all DCT coefficients lie in [(d-1/2)q, (d+1/2)q] (SURVEY.md §8c ii)."""
    planes = make_case(64, 48, "444", 10, seed=5, y_only=True)
    out, _ = oracle.oracle_compute(planes, 0.3, [0.001], 6)
    p = planes[0]
    blocks = out[0].reshape(p.h // 8, 8, p.w // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    coefs = oracle.dct_blocks(blocks, inverse=False).astype(np.float64)
    d = p.data.reshape(-1, 64).astype(np.float64)
    q = p.quant_table.astype(np.float64)
    assert (coefs >= (d - 0.5) * q - 1e-3).all() and (coefs <= (d + 0.5) * q + 1e-3).all()


def test_synth_is_band_invariant():
    """a band generated on its own equals the same rows of the whole image (row-tiled runs)."""
    from jpeg2png_amd import synth
    whole = synth.make_planes(96, 128, "444", 10, seed=9, y_only=True)[0]
    band = synth.make_planes(96, 128, "444", 10, seed=9, y_only=True, rows=(64, 128))[0]
    assert band.h == 64 and band.w == 96
    assert np.array_equal(band.data, whole.data.reshape(16, -1)[8:].reshape(-1))


def test_oracle_matches_reference_on_sweep_cases():
    """the restatement against the compiled reference on the randomised case stream the GPU tests use
    (small cases only: the oracle is a slow scalar program)"""
    from oracle import bindings
    if not bindings.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    from sweep_cases import cases
    done = 0
    for cs in cases(11, 60):
        if cs.W * cs.H * cs.iterations > 1.5e6:
            continue
        planes = cs.planes()
        for p in planes:
            p.fdata = bindings.decode_plane(p)
        want, want_log, _ = bindings.ref_compute(planes, cs.weight, cs.pweights, cs.iterations, log=cs.log)
        got, got_log = bindings.oracle_compute(planes, cs.weight, cs.pweights, cs.iterations, log=cs.log)
        for c in range(len(planes)):
            assert bit_equal(got[c], want[c]), cs.describe()
        if cs.log and cs.iterations:
            np.testing.assert_allclose(got_log[:, 1:], want_log[:, 1:], rtol=1e-9, atol=2e-6)
        done += 1
    assert done >= 15
