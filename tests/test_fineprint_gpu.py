"""The fine print of the parity statement, made executable (INTEGRATION.md "Where the results may differ"):
documented divergences from the compiled reference and the error contract of the non-dying entry point."""
import copy
import ctypes

import numpy as np
import pytest

from conftest import bit_equal, make_case

pytestmark = pytest.mark.gpu


def test_quant_entries_above_32767_follow_the_c_build(lib, oracle):
    """quantisation tables with entries >= 32768 (16-bit-precision JPEGs only): the reference's own two builds disagree
    there — its SSE2 routines convert the uint16 table with the SIGNED _mm_cvtpi16_ps (compute_simd_step.c:17,160), its
    C routines with (float)q (compute.c:47,326).  The kernels and the oracle's restatement follow the C build;
    oracle/_ref is the SIMD build (the reference's default), so THIS is the one input class where the parity statements
    are against the restatement, not against oracle/_ref — and the test says so by checking that the two CPU sides differ."""
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    p = synth.make_planes(64, 48, "444", 10, seed=5, y_only=True)[0]
    q = p.quant_table.copy().astype(np.uint16)
    q[0], q[5], q[63] = 33000, 40000, 65535
    d = p.data.reshape(-1, 64).copy()
    rng = np.random.default_rng(1)
    for k in (0, 5, 63):
        d[:, k] = rng.integers(-1, 2, d.shape[0])
    plane = synth.Plane(p.w, p.h, 1, 1, d.reshape(-1), q)
    plane.fdata = oracle.decode_plane(plane)
    assert bit_equal(j.decode_plane(plane), plane.fdata)
    for pweight in (0.001, 0.0):
        want, want_log = oracle.oracle_compute([plane], 0.3, [pweight], 6, log=True)
        got = copy.deepcopy([plane])
        got_log = j.compute(got, 0.3, [pweight], 6, log=True)
        assert np.isfinite(got[0].fdata).all()
        assert bit_equal(got[0].fdata, want[0]), f"pweight {pweight}: the C build's result is the documented one"
        np.testing.assert_allclose(got_log, want_log, rtol=1e-9, atol=1e-9)
        if oracle.have_ref():
            ref, _, _ = oracle.ref_compute([copy.deepcopy(plane)], 0.3, [pweight], 6)
            assert not bit_equal(ref[0], want[0]), "the reference's SIMD build was expected to differ from its C build here"


@pytest.mark.timeout(120)
@pytest.mark.parametrize("W,H", [(96, 64), (1600, 1200)])
def test_a_nan_pixel_neither_hangs_the_run_nor_survives_the_projection(lib, oracle, W, H):
    """non-finite input is outside the parity statement (the reference's clamp hands a NaN through, compute.c:327-329,
    and its image is NaN everywhere two iterations later); what IS promised: the run returns — the norm fold's parity
    spin forces the sign bit of every partial precisely so that a NaN sum cannot make the reader wait for ever
    (j2p_kernels.hip.h, fold_tile_row) — and the v_med3_f32 clamp returns the interval's lower end for a NaN coefficient,
    so the plane handed back is finite and every coefficient of it lies in its quantisation interval.  Small canvas: the
    folded reduction; 1.9 Mpixel: fold + the in-projection tree on a filled chip."""
    import jpeg2png_amd as j
    planes = make_case(W, H, "444", 10, seed=6, y_only=True)
    planes[0].fdata = planes[0].fdata.copy()
    planes[0].fdata[H // 2 + 3, W // 2 + 5] = np.nan
    got = copy.deepcopy(planes)
    j.compute(got, 0.3, [0.001], 5)
    out = got[0].fdata
    assert np.isfinite(out).all()
    # projection property (compute.c:323-331): DCT coefficients inside [(d - 1/2) q, (d + 1/2) q] up to float rounding
    blocks = out.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    coef = oracle.dct_blocks(blocks)
    d = planes[0].data.reshape(-1, 64).astype(np.float64)
    q = planes[0].quant_table.astype(np.float64)[None, :]
    assert (np.abs(coef - d * q) <= 0.5 * q * (1 + 1e-4) + 1e-3).all()


class _Coef(ctypes.Structure):
    _fields_ = [("h", ctypes.c_uint), ("w", ctypes.c_uint), ("h_samp", ctypes.c_uint), ("w_samp", ctypes.c_uint),
                ("data", ctypes.c_void_p), ("fdata", ctypes.c_void_p), ("quant_table", ctypes.c_uint16 * 64)]


def _libc_coefs(planes):
    libc = ctypes.CDLL(None)
    libc.aligned_alloc.restype = ctypes.c_void_p
    libc.aligned_alloc.argtypes = [ctypes.c_size_t, ctypes.c_size_t]
    libc.malloc.restype = ctypes.c_void_p
    libc.malloc.argtypes = [ctypes.c_size_t]
    libc.free.argtypes = [ctypes.c_void_p]
    coefs = (_Coef * len(planes))()
    for c, p in enumerate(planes):
        d = np.ascontiguousarray(p.data, dtype=np.int16)
        f = np.ascontiguousarray(p.fdata, dtype=np.float32)
        coefs[c].w, coefs[c].h, coefs[c].w_samp, coefs[c].h_samp = p.w, p.h, p.w_samp, p.h_samp
        coefs[c].data = libc.malloc(d.nbytes)
        coefs[c].fdata = libc.aligned_alloc(16, (f.nbytes + 15) & ~15)
        ctypes.memmove(coefs[c].data, d.ctypes.data, d.nbytes)
        ctypes.memmove(coefs[c].fdata, f.ctypes.data, f.nbytes)
        for k, qv in enumerate(np.asarray(p.quant_table, dtype=np.uint16).reshape(64)):
            coefs[c].quant_table[k] = int(qv)
    return libc, coefs


@pytest.mark.parametrize("tiled", [False, True])
def test_j2p_compute_keeps_the_callers_planes_when_the_solve_fails(lib, tiled):
    """j2p_compute() promises an error code instead of exit() (include/jpeg2png_amd_compute.h): a failing call must hand
    the caller's planes back untouched — pointer, size and content — so that the caller can retry (on another device,
    say); the retry then gives the right answer.  The failure is injected behind a successful create
    (j2p_debug_fail_run_after): the upload has happened, the output planes exist, nothing may have been released.
    And what a successful call does with the planes (compute.c:304-305, 455-461 as the caller sees them): a plane that
    already has the canvas's size — the luma of this unpadded 4:2:0 image — keeps its buffer, the result is downloaded
    into it; planes that must grow (chroma) are replaced."""
    import jpeg2png_amd as j
    planes = make_case(208, 176, "420", 10, seed=31)          # (a multiple of 16 each way: chroma pads no further than luma)
    want = copy.deepcopy(planes)
    j.compute(want, 0.3, [0.001] * 3, 7)
    libc, coefs = _libc_coefs(planes)
    before = [(coefs[c].fdata, coefs[c].w, coefs[c].h) for c in range(3)]
    pw = (ctypes.c_float * 3)(0.001, 0.001, 0.001)
    devs = (ctypes.c_int * 2)(0, 0)

    def call():
        if tiled:
            lib.j2p_compute_tiled.argtypes = [ctypes.c_uint, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_uint]
            return lib.j2p_compute_tiled(2, devs, 3, coefs, None, None, 0.3, pw, 7)
        lib.j2p_compute.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float,
                                    ctypes.c_void_p, ctypes.c_uint]
        return lib.j2p_compute(0, 3, coefs, None, None, 0.3, pw, 7)
    lib.j2p_debug_fail_run_after.argtypes = [ctypes.c_int]
    lib.j2p_debug_fail_run_after.restype = None
    lib.j2p_debug_fail_run_after(1)
    try:
        rc = call()
    finally:
        lib.j2p_debug_fail_run_after(0)
    assert rc == -3 and b"injected failure" in lib.j2p_last_error()
    for c in range(3):
        assert (coefs[c].fdata, coefs[c].w, coefs[c].h) == before[c], f"channel {c}: the inputs changed hands on an error return"
        a = np.ctypeslib.as_array(ctypes.cast(coefs[c].fdata, ctypes.POINTER(ctypes.c_float)), shape=(planes[c].h, planes[c].w))
        assert bit_equal(a, planes[c].fdata), f"channel {c}: input plane damaged"
    # the retry
    assert call() == 0, lib.j2p_last_error()
    assert coefs[0].fdata == before[0][0], "a full-resolution plane of the canvas's size should have been reused"
    for c in range(3):
        assert (coefs[c].w, coefs[c].h) == (want[0].fdata.shape[1], want[0].fdata.shape[0])
        a = np.ctypeslib.as_array(ctypes.cast(coefs[c].fdata, ctypes.POINTER(ctypes.c_float)), shape=(coefs[c].h, coefs[c].w))
        assert bit_equal(a, want[c].fdata), f"channel {c} after the retry"
        libc.free(coefs[c].fdata)
        libc.free(coefs[c].data)


def test_compute_timing_adds_up(lib):
    """j2p_compute_timing(): the split bench.py's host_to_host object reports — present after a successful call, its
    parts sum to the total, and the total is what the caller's clock saw"""
    import jpeg2png_amd as j
    planes = make_case(640, 480, "444", 10, seed=32, y_only=True)
    splits = []
    _, secs = j.compute_c(planes, 0.3, [0.001], 20, repeat=2, splits=splits)
    assert len(splits) == 2
    for sp, wall in zip(splits, secs):
        parts = sp["create_ms"] + sp["issue_ms"] + sp["wait_ms"] + sp["download_ms"] + sp["destroy_ms"]
        assert abs(parts - sp["total_ms"]) < 0.05
        assert sp["total_ms"] <= wall * 1e3 + 0.05 and sp["total_ms"] >= wall * 1e3 * 0.5
        assert all(v >= 0 for v in sp.values())
