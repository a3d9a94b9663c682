"""Parity at the parameters BASELINE.json's `configs` state, one test per configuration, against the UNMODIFIED
reference (oracle/_ref: compute.c compiled from /root/reference) on identical `struct coef` inputs.

Bar (north_star): per-plane PSNR >= 80 dB, peak 255 — asserted hard.  Expected AND asserted, at every size:
BIT-IDENTICAL planes.  The one place where that could fail without a defect: ||g|| is a double sum whose (fixed,
GPU-count invariant) tree order differs from the reference's sequential order (compute.c:200-207), and the two can
land on different sides of a float rounding boundary inside sqrtf((float)sum) with probability ~1e-5 per iteration
at 16 Mpixel (SURVEY.md §7 hard part 2) — not observed so far.  Should it ever happen on the large single planes
(configs[2], configs[3]) the test FAILS, so that the record and README.md cannot disagree; J2P_ALLOW_NORM_FLIP=1 in
the environment turns that one failure (never the 80 dB bar) into a reported WARNING line.
"""
import copy
import os
import threading
import time
import warnings

import numpy as np
import pytest

from conftest import band_devices, bit_equal, parity_note, psnr

pytestmark = pytest.mark.gpu

PSNR_BAR_DB = 80.0
WEIGHT, PWEIGHT = 0.3, 0.001          # jpeg2png.c:22-23
# large single planes: a bit mismatch above 80 dB fails the test unless this is set (see the module docstring)
ALLOW_NORM_FLIP = os.environ.get("J2P_ALLOW_NORM_FLIP", "") == "1"


def _need_ref(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")


def check_planes(name, got, want, strict=True, against="the reference"):
    """hard: 80 dB.  Bit-identity: hard when `strict`, a reported warning otherwise.  Either way the outcome goes
    into the run's "parity" summary (conftest.parity_note)."""
    identical = True
    for c, (g, w) in enumerate(zip(got, want)):
        db = psnr(g, w)
        assert db >= PSNR_BAR_DB, f"{name} channel {c}: PSNR {db:.1f} dB vs {against}"
        if not bit_equal(g, w):
            identical = False
            bad = int(np.count_nonzero(np.ascontiguousarray(g).view(np.uint32) != np.ascontiguousarray(w).view(np.uint32)))
            msg = (f"{name} channel {c}: {db:.1f} dB but {bad} of {g.size} pixels differ in their bits "
                   "(a one-ulp flip of ||g||?)")
            parity_note("WARNING " + msg)
            if strict:
                raise AssertionError(msg)
            warnings.warn(msg)
    if identical:
        parity_note(f"{name}: bit-identical to {against} ({len(got)} plane{'s' if len(got) != 1 else ''}, "
                    f"{got[0].shape[1]}x{got[0].shape[0]})")


def check_log(got_log, want_log):
    # tv, tv2, prob_dist against the reference's CSV (%f: 6 decimals)
    np.testing.assert_allclose(got_log[:, 1:], want_log[:, 1:], rtol=1e-9, atol=2e-6)


def test_config0_512x512_420_q10_joint_i50(lib, oracle):
    """configs[0]: 512x512 4:2:0 JPEG Q=10, -i 50, joint (the call jpeg2png.c:144 makes)"""
    _need_ref(oracle)
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    planes = synth.make_planes(512, 512, "420", 10, seed=1234 + 1)
    for p in planes:
        p.fdata = oracle.decode_plane(p)
    want, want_log, _ = oracle.ref_compute(planes, WEIGHT, [PWEIGHT] * 3, 50, log=True)
    got = copy.deepcopy(planes)
    got_log = j.compute(got, WEIGHT, [PWEIGHT] * 3, 50, log=True)
    check_planes("configs[0]", [p.fdata for p in got], want)
    check_log(got_log, want_log)


def test_config1_1080p_444_q10_separate_on_three_streams_i100(lib, oracle):
    """configs[1]: 1920x1080 4:4:4 Q=10, -i 100, `-s`: three compute(1, ...) calls (jpeg2png.c:147-152) with the
    CLI's default weights (0.3, 0, 0), here in flight at the same time on three streams of one GPU"""
    _need_ref(oracle)
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    planes = synth.make_planes(1920, 1080, "444", 10, seed=1234 + 2)
    for p in planes:
        p.fdata = oracle.decode_plane(p)
    weights = [WEIGHT, 0.0, 0.0]                                     # jpeg2png.c:206
    wants = [oracle.ref_compute([planes[c]], weights[c], [PWEIGHT], 100)[0][0] for c in range(3)]
    solvers = [j.Solver([planes[c]], weights[c], [PWEIGHT], 100) for c in range(3)]
    errs = []

    def work(s):
        try:
            s.run(100)
            s.sync()
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(s,)) for s in solvers]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    got = [s.download(0) for s in solvers]
    for s in solvers:
        s.close()
    check_planes("configs[1]", got, wants)


def test_config2_4096x4096_y_q10_i500_the_bench_workload(lib, oracle):
    """configs[2]: 4096x4096 Y-only Q=10 -i 500 — exactly what bench.py times (same seed).  ~90 s of one CPU
    core inside the reference's compute()."""
    _need_ref(oracle)
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    planes = synth.make_planes(4096, 4096, "444", 10, seed=1234 + 3, y_only=True)
    for p in planes:
        p.fdata = j.decode_plane(p)
    want, _, secs = oracle.ref_compute(planes, WEIGHT, [PWEIGHT], 500)
    got = copy.deepcopy(planes)
    j.compute(got, WEIGHT, [PWEIGHT], 500)
    check_planes("configs[2]", [got[0].fdata], want, strict=not ALLOW_NORM_FLIP)
    print(f"configs[2]: reference {secs:.1f} s inside compute()")


def test_config3_16384_wide_bands_match_whole_and_reference(lib, oracle):
    """configs[3] is a 16384-wide plane cut into 2048-row bands.  (a) band-split invariance at that width: a
    16384x2048 canvas as 8 bands of 256 rows through the C row-tiling engine (j2p_tiled: one host thread per
    band, event-ordered exchanges; all bands on this box's one GPU) vs the whole-canvas solver, 5 iterations,
    bitwise; (b) a 16384x1024 plane vs the compiled reference, 5 iterations."""
    _need_ref(oracle)
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    W, its = 16384, 5
    # (b)
    planes = synth.make_planes(W, 1024, "444", 10, seed=1234 + 4, y_only=True)
    for p in planes:
        p.fdata = j.decode_plane(p)
    want, _, _ = oracle.ref_compute(planes, WEIGHT, [PWEIGHT], its)
    got = copy.deepcopy(planes)
    j.compute(got, WEIGHT, [PWEIGHT], its)
    check_planes("configs[3] 16384x1024 vs reference", [got[0].fdata], want)
    # (a)
    H = 2048
    planes = synth.make_planes(W, H, "444", 10, seed=1234 + 4, y_only=True)
    for p in planes:
        p.fdata = j.decode_plane(p)
    with j.Solver(planes, WEIGHT, [PWEIGHT], its) as s:
        s.run(its)
        whole = s.download(0)
    with j.TiledSolver(planes, WEIGHT, [PWEIGHT], its, devices=band_devices(8)) as t:
        assert [b[1:] for b in t.bands()] == [(r, r + 256) for r in range(0, H, 256)]
        t.run(its)
        banded = t.download(0)
    assert bit_equal(banded, whole), "16384-wide canvas: 8 bands differ from the whole-canvas solve"


def test_config4_1080p_420_q50_joint_i100(lib, oracle):
    """configs[4]: one image of the batch — 1080p 4:2:0 Q=50, -i 100, joint (padded 1920x1088 canvas: the luma
    plane goes through the resampling path with a 1x1 footprint)"""
    _need_ref(oracle)
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    planes = synth.make_planes(1920, 1080, "420", 50, seed=1234 + 5)
    for p in planes:
        p.fdata = oracle.decode_plane(p)
    want, want_log, _ = oracle.ref_compute(planes, WEIGHT, [PWEIGHT] * 3, 100, log=True)
    got = copy.deepcopy(planes)
    got_log = j.compute(got, WEIGHT, [PWEIGHT] * 3, 100, log=True)
    check_planes("configs[4]", [p.fdata for p in got], want)
    check_log(got_log, want_log)


def test_config3_full_size_i100_vs_reference_whole_and_8_bands(lib, oracle, config3_reference):
    """configs[3] AT ITS STATED PARAMETERS against the UNMODIFIED reference: 16384x16384 Y-only Q10, `-i 100` (the
    iteration count enters the step size, compute.c:443, so no shorter run is the same solve).  The reference's
    compute() on that plane takes ~5 minutes of one CPU core and ~9 GiB of host memory: it was started in a worker
    thread when the session began (conftest.config3_reference; ctypes releases the GIL) and is joined here, the last
    test of the run.  Compared with it, bitwise: (a) the whole-canvas solver through compute()'s path, (b) the C row
    tiling, 8 bands of 2048 rows (j2p_tiled; the bands share this box's one GPU) — the loop compute.c:427-453 on the
    16384x16384 geometry; (b) against (a) too, CSV rows included."""
    _need_ref(oracle)
    import jpeg2png_amd as j
    its = 100
    plane, want, ref_seconds, waited, no_reference = config3_reference()
    with j.Solver([plane], WEIGHT, [PWEIGHT], its) as s:
        whole_rows = s.run(its, log=True)
        whole = s.download(0)
    if want is None:
        # no live reference run on this lease (conftest: memory / time budget): the same plane, the same solve, compared
        # through the digest of the reference's plane that tests/golden/bench_digests.json holds for it
        import hashlib
        import json
        entry = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bench_digests.json")))["configs[3] N=8"]
        assert (entry["W"], entry["H"], entry["iterations"], entry["seed"]) == (16384, 16384, its, 1234 + 4)
        got = hashlib.blake2b(np.ascontiguousarray(whole), digest_size=16).hexdigest()
        parity_note(f"configs[3] 16384x16384 -i 100: NO LIVE REFERENCE RUN ({no_reference}); compared with the reference's plane by its "
                    f"digest in tests/golden/bench_digests.json instead: {'bit-identical' if got == entry['digest'] else 'DIFFERENT'}")
        assert got == entry["digest"] or ALLOW_NORM_FLIP, "16384x16384 -i 100, whole canvas: not the reference's plane (digest)"
        want = [whole]            # (the bands are held against the whole-canvas solve below, which the digest has just vouched for)
    else:
        check_planes("configs[3] 16384x16384 -i 100, whole canvas", [whole], want, strict=not ALLOW_NORM_FLIP)
    j.load_library().j2p_pool_trim()
    with j.TiledSolver([plane], WEIGHT, [PWEIGHT], its, devices=band_devices(8)) as t:
        assert [b[1:] for b in t.bands()] == [(r, r + 2048) for r in range(0, 16384, 2048)]
        banded_rows = t.run(its, log=True)
        banded = t.download(0)
        cpu = t.host_cpu_seconds()
    check_planes("configs[3] 16384x16384 -i 100, 8 x 2048-row bands", [banded], want, strict=not ALLOW_NORM_FLIP)
    assert bit_equal(banded, whole), "16384x16384 -i 100: 8 bands differ from the whole-canvas solve"
    np.testing.assert_allclose(banded_rows, whole_rows, rtol=1e-9, atol=1e-9)
    assert np.isfinite(banded_rows).all()
    parity_note(f"configs[3] 16384x16384 -i 100: 8 bands == whole canvas bitwise, CSV rows equal; band threads used {cpu:.3f} s of "
                f"host CPU; the reference spent {ref_seconds:.0f} s inside compute() (in the background; this test waited {waited:.0f} s)")


def test_config4_batch_of_256_images_vs_reference(lib, oracle):
    """configs[4] AS STATED: a batch of 256 x 1080p 4:2:0 Q50 `-i 100` joint through the C batch engine (j2p_batch: worker
    slots, pooled arenas, overlapping upload / solve / download — the file loop jpeg2png.c:330-337), float canvas planes
    out.  Images 0, 255 and one drawn at random against the reference's compute() on the same struct coef; every other
    image against the first occurrence of its content (8 distinct images, cycled).  At most 32 results are held at a time."""
    _need_ref(oracle)
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    n, its, ndistinct, window = 256, 100, 8, 32
    distinct = [synth.make_planes(1920, 1080, "420", 50, seed=1234 + 5 + k) for k in range(ndistinct)]
    pick = int(np.random.default_rng(20260925).integers(1, n - 1))
    against_ref = {0, n - 1, pick}
    first = {}                      # content index -> planes of its first occurrence
    kept = {}                       # image index -> planes, for the reference comparison
    t0 = time.perf_counter()
    with j.Batch(devices=band_devices(j.device_count()), slots_per_device=8) as b:      # every GPU that is there
        tickets = {}

        def collect(i):
            out = b.wait(tickets.pop(i))
            k = i % ndistinct
            if k not in first:
                first[k] = out
            else:
                for c in range(3):
                    assert bit_equal(out[c], first[k][c]), f"image {i} channel {c} differs from image {k} (same content)"
            if i in against_ref:
                kept[i] = out
        for i in range(n):
            tickets[i] = b.submit(distinct[i % ndistinct], WEIGHT, [PWEIGHT] * 3, its)
            if i >= window:
                collect(i - window)
        for i in range(n - window, n):
            collect(i)
    secs = time.perf_counter() - t0
    for i in sorted(against_ref):
        planes = copy.deepcopy(distinct[i % ndistinct])
        for p in planes:
            p.fdata = oracle.decode_plane(p)
        want, _, _ = oracle.ref_compute(planes, WEIGHT, [PWEIGHT] * 3, its)
        check_planes(f"configs[4] batch of {n}, image {i}", kept[i], want)
    parity_note(f"configs[4] batch of {n} x 1080p 4:2:0 Q50 -i 100: every image bit-identical to the first occurrence of its "
                f"content ({ndistinct} distinct); {n / secs:.0f} images/s host-to-host incl. the comparisons")
