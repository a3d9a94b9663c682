"""The image-batch engine (j2p_batch_*, jpeg2png_amd/csrc/j2p_batch.hip): jobs on several slots of one GPU must
give exactly what one compute() call per image gives — float planes bit for bit, RGB samples byte for byte —
whatever the interleaving, and the pool must recycle device memory between images."""
import copy

import numpy as np
import pytest

from conftest import band_devices, bit_equal, make_case

pytestmark = pytest.mark.gpu


def test_batch_planes_equal_individual_computes(lib):
    import jpeg2png_amd as j
    cases = [make_case(120 + 16 * i, 88 + 8 * i, ["420", "444", "422"][i % 3], 10 + 5 * i, seed=60 + i) for i in range(7)]
    its = 14
    with j.Batch(devices=[0], slots_per_device=3) as b:
        # device-side decode (fdata = None) for half of the jobs, host-decoded planes for the others
        jobs = []
        for i, planes in enumerate(cases):
            sub = copy.deepcopy(planes)
            if i % 2:
                for p in sub:
                    p.fdata = None
            jobs.append(b.submit(sub, 0.3, [0.001] * 3, its))
        outs = [b.wait(t) for t in jobs]
    for planes, out in zip(cases, outs):
        ref = copy.deepcopy(planes)
        j.compute(ref, 0.3, [0.001] * 3, its)
        for c in range(3):
            assert bit_equal(out[c], ref[c].fdata)


def test_batch_separate_components_and_rgb(lib):
    """-s semantics (three compute(1, ...) with their own weights / iteration counts, jpeg2png.c:147-152) and the
    RGB output path against three single solvers + j2p_planes_to_rgb"""
    import ctypes
    import jpeg2png_amd as j
    planes = make_case(200, 136, "420", 10, seed=71)
    weights, its = [0.3, 0.1, 0.0], [20, 9, 5]
    with j.Batch(devices=[0], slots_per_device=2) as b:
        t8 = b.submit(planes, weights, [0.001] * 3, its, separate=True, width=200, height=136, bits=8)
        t16 = b.submit(planes, weights, [0.001] * 3, its, separate=True, width=200, height=136, bits=16)
        tf = b.submit(planes, weights, [0.001] * 3, its, separate=True)
        rgb8, rgb16, fl = b.wait(t8), b.wait(t16), b.wait(tf)
    solvers = [j.Solver([planes[c]], weights[c], [0.001], its[c]) for c in range(3)]
    for c, s in enumerate(solvers):
        s.run(its[c])
        assert bit_equal(fl[c], s.download(0))

    class Ref(ctypes.Structure):
        _fields_ = [("solver", ctypes.c_void_p), ("channel", ctypes.c_uint)]
    lib.j2p_planes_to_rgb.argtypes = [ctypes.POINTER(Ref), ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p]
    for bits, got in ((8, rgb8), (16, rgb16)):
        refs = (Ref * 3)(*[Ref(s._h, 0) for s in solvers])
        want = np.empty_like(got)
        assert lib.j2p_planes_to_rgb(refs, 200, 136, bits, want.ctypes.data) == 0
        assert np.array_equal(got, want)
    for s in solvers:
        s.close()


def test_batch_refuses_an_output_array_of_the_wrong_kind(lib):
    """submit(out=...) hands the C side a bare pointer it fills with height x width x 3 samples of bits / 8 bytes: an array
    of another element size (a uint8 ring reused for a 16-bit job would be overrun by a factor of two), shape, layout or a
    read-only one is refused before anything is queued — and the right one is filled"""
    import jpeg2png_amd as j
    planes = make_case(64, 48, "420", 20, seed=4)
    with j.Batch(devices=[0], slots_per_device=1) as b:
        for bits, arr in ((16, np.empty((48, 64, 3), np.uint8)), (8, np.empty((48, 64, 3), np.float32)),
                          (8, np.empty((48, 64, 4), np.uint8)), (8, np.empty((64, 48, 3), np.uint8).transpose(1, 0, 2)),
                          (12, np.empty((48, 64, 3), np.uint8))):
            with pytest.raises(j.J2PError):
                b.submit(planes, 0.3, [0.001] * 3, 3, width=64, height=48, bits=bits, out=arr)
        ro = np.empty((48, 64, 3), np.uint8)
        ro.flags.writeable = False
        with pytest.raises(j.J2PError):
            b.submit(planes, 0.3, [0.001] * 3, 3, width=64, height=48, bits=8, out=ro)
        good8, good16 = np.zeros((48, 64, 3), np.uint8), np.zeros((48, 64, 3), ">u2")
        t8 = b.submit(planes, 0.3, [0.001] * 3, 3, width=64, height=48, bits=8, out=good8)
        t16 = b.submit(planes, 0.3, [0.001] * 3, 3, width=64, height=48, bits=16, out=good16)
        assert b.wait(t8) is good8 and b.wait(t16) is good16
        assert good8.any() and good16.any()


@pytest.mark.parametrize("tile", [False, True])
def test_progress_ticks_follow_the_clock(lib, tile):
    """the CLI's progress bar is fed through j2p_job.on_progress: like the reference's once-per-iteration tick
    (compute.c:449-452) it must move WHILE the solve runs — one iteration per round trip at first, then a sixth of what is
    done, never more than ~50 ms worth (j2p_batch.hip: next_chunk; compute_host.c does the same for compute()) — not in a
    few bursts of 32; one solver and the row tiling"""
    import time
    import jpeg2png_amd as j
    planes = make_case(160, 144, "420", 10, seed=9)
    its, ticks = 50, []
    with j.Batch(devices=band_devices(2 if tile else 1), slots_per_device=1) as b:
        t = b.submit(planes, 0.3, [0.001] * 3, its, tile=tile, tile_min_band_pixels=0,
                     on_progress=lambda n: ticks.append((time.perf_counter(), n)))
        out = b.wait(t)
    assert sum(n for _, n in ticks) == its
    assert len(ticks) >= 20 and ticks[0][1] == 1 and max(n for _, n in ticks) <= 9
    ref = copy.deepcopy(planes)
    j.compute(ref, 0.3, [0.001] * 3, its)
    for c in range(3):
        assert bit_equal(out[c], ref[c].fdata)              # chunking changes nothing about the result


def test_batch_reports_a_bad_job_and_carries_on(lib):
    import jpeg2png_amd as j
    good = make_case(64, 48, "444", 20, seed=3)
    bad = copy.deepcopy(good)
    bad[1].quant_table = np.zeros(64, np.uint16)               # jpeg.c:41-45
    with j.Batch(devices=[0], slots_per_device=2) as b:
        tb = b.submit(bad, 0.3, [0.001] * 3, 4)
        tg = b.submit(good, 0.3, [0.001] * 3, 4)
        with pytest.raises(j.J2PError, match="quantization table"):
            b.wait(tb)
        out = b.wait(tg)
    ref = copy.deepcopy(good)
    j.compute(ref, 0.3, [0.001] * 3, 4)
    assert bit_equal(out[0], ref[0].fdata)


def test_pool_recycles_arenas(lib):
    """after a first solver of a given size, the next ones on the device get its memory back: same address"""
    import jpeg2png_amd as j
    planes = make_case(256, 192, "420", 10, seed=8)
    lib.j2p_pool_trim()
    s1 = j.Solver(planes, 0.3, [0.001] * 3, 4)
    a1 = s1.plane_ptr(0)
    s1.close()
    s2 = j.Solver(planes, 0.3, [0.001] * 3, 4)
    a2 = s2.plane_ptr(0)
    s2.run(4)
    out = [s2.download(c) for c in range(3)]
    s2.close()
    assert a1 == a2
    ref = copy.deepcopy(planes)
    j.compute(ref, 0.3, [0.001] * 3, 4)
    for c in range(3):
        assert bit_equal(out[c], ref[c].fdata)          # a recycled (dirty) arena changes nothing
    lib.j2p_pool_trim()


@pytest.mark.parametrize("separate", [False, True])
def test_tiled_job_equals_single_solver_job(lib, separate, monkeypatch):
    """j2p_job.tile: one image over all the batch's devices (here device 0 three times): float planes and RGB bytes
    equal the untiled job's; a canvas too short to tile falls back to the single-solver path.  (The library tiles only
    images with at least 2 Mpixel per band; the gate is lowered here so that a small image exercises the path.)"""
    import jpeg2png_amd as j
    planes = make_case(152, 296, "420", 10, seed=83)
    weights, its = ([0.3, 0.1, 0.0], [12, 7, 5]) if separate else (0.3, 12)
    with j.Batch(devices=band_devices(3), slots_per_device=1) as b:
        a_f = b.wait(b.submit(planes, weights, [0.001] * 3, its, separate=separate))
        a_rgb = b.wait(b.submit(planes, weights, [0.001] * 3, its, separate=separate, width=150, height=290, bits=8))
        t_f = b.wait(b.submit(planes, weights, [0.001] * 3, its, separate=separate, tile=True, tile_min_band_pixels=0))
        t_rgb = b.wait(b.submit(planes, weights, [0.001] * 3, its, separate=separate, width=150, height=290, bits=8, tile=True, tile_min_band_pixels=0))
        t_rgb16 = b.wait(b.submit(planes, weights, [0.001] * 3, its, separate=separate, width=150, height=290, bits=16, tile=True, tile_min_band_pixels=0))
        a_rgb16 = b.wait(b.submit(planes, weights, [0.001] * 3, its, separate=separate, width=150, height=290, bits=16))
        short = make_case(64, 48, "444", 20, seed=3)
        s_t = b.wait(b.submit(short, 0.3, [0.001] * 3, 4, tile=True, tile_min_band_pixels=0))
        s_a = b.wait(b.submit(short, 0.3, [0.001] * 3, 4))
    for c in range(3):
        assert bit_equal(t_f[c], a_f[c]), f"channel {c}"
        assert bit_equal(s_t[c], s_a[c])
    assert np.array_equal(t_rgb, a_rgb)
    assert np.array_equal(t_rgb16, a_rgb16)


def test_tile_gate_and_device_shares(lib, monkeypatch):
    """a job's share of the batch's devices (tile_first / tile_count: a few files, fewer than GPUs) and the pixel gate:
    below 2 Mpixel per band the image is solved on one GPU of its share — same bits every way; a share outside the
    batch's device list and a bad output description are rejected before anything runs"""
    import jpeg2png_amd as j
    planes = make_case(152, 296, "420", 10, seed=84)
    with j.Batch(devices=band_devices(4), slots_per_device=1) as b:
        plain = b.wait(b.submit(planes, 0.3, [0.001] * 3, 6))
        gated = b.wait(b.submit(planes, 0.3, [0.001] * 3, 6, tile=True, tile_devices=(2, 2)))       # too small: untiled
        share = b.wait(b.submit(planes, 0.3, [0.001] * 3, 6, tile=True, tile_devices=(1, 2), tile_min_band_pixels=0))       # two bands
        with pytest.raises(j.J2PError, match="tile devices"):
            b.wait(b.submit(planes, 0.3, [0.001] * 3, 6, tile=True, tile_devices=(3, 2), tile_min_band_pixels=0))
        with pytest.raises(j.J2PError, match="out_bits"):
            b.wait(b.submit(planes, 0.3, [0.001] * 3, 6, tile=True, width=150, height=290, bits=12))
    for c in range(3):
        assert bit_equal(gated[c], plain[c]) and bit_equal(share[c], plain[c])
