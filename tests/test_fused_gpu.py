"""EXPERIMENTS BUILD ONLY — measured slower than two launches at every size and dropped from the release library
(profiles/r05_single_launch.jsonl); kept bit-checked here so that the measurement stays reproducible.
The single-launch iteration (k_iterate, J2P_OPT_FUSE): projection(k) and gradient(k + 1) of one full-resolution channel
in ONE grid, gradient workgroups waiting on per-block-row counters for the rows they read (reference loop:
compute.c:430-448; ||g||, compute.c:209-211, is the only device-wide dependency left between launches).  Another schedule
of the same arithmetic: the planes must be the compiled reference's bit for bit, whatever the canvas shape, however the
iterations are cut into run() calls, with the chip shared by other solvers."""
import copy
import threading

import numpy as np
import pytest

from conftest import bit_equal, make_case, parity_note

pytestmark = pytest.mark.gpu


def reference(oracle, planes, weight, pw, its):
    if oracle.have_ref():
        return oracle.ref_compute(copy.deepcopy(planes), weight, [pw], its)[0]
    return oracle.oracle_compute(planes, weight, [pw], its)[0]


@pytest.mark.parametrize("W,H,weight,pw", [(8, 8, 0.3, 0.001), (64, 64, 0.3, 0.001), (72, 40, 0.3, 0.001), (200, 136, 0.0, 0.001),
                                            (264, 410, 0.3, 0.0), (1000, 600, 0.3, 0.001), (1920, 1080, 0.3, 0.001),
                                            (2048, 2048, 0.3, 0.001), (4096, 1024, 0.0, 0.001)])
def test_single_launch_iteration_matches_the_reference(exp_lib, oracle, W, H, weight, pw):
    """ragged widths (strips that stick out of the canvas take the projection's generic path), tiny canvases (one workgroup of
    each kind), TV-only, prob term off, the sizes the schedule is meant for — iterations in uneven run() calls, so that runs
    open and close with the plain kernels at every parity"""
    import jpeg2png_amd as j
    planes = make_case(W, H, "444", 10, seed=5 + W, y_only=True)
    its = 13
    want = reference(oracle, planes, weight, pw, its)
    for fuse in (1, 0):
        with j.Solver(planes, weight, [pw], its) as s:
            s.debug_option(j.J2P_OPT_FUSE, fuse)
            assert s.launches_per_iteration() == 1 or not fuse
            for n in (3, 1, 2, 5, 2):
                s.run(n)
            got = s.download(0)
            assert bit_equal(got, want[0]), f"{W}x{H} fuse {fuse}"
            s.reset()
            s.run(its)
            assert bit_equal(s.download(0), want[0]), f"{W}x{H} fuse {fuse}, after reset"
    parity_note(f"single-launch iteration {W}x{H} weight {weight} pweight {pw}: bit-identical to the reference")


def test_it_is_never_the_librarys_own_choice(lib):
    """release library: two or three launches per iteration whatever the plane, and the switch says where the schedule lives"""
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    y = synth.make_planes(1920, 1080, "444", 10, seed=3, y_only=True)
    with j.Solver(y, 0.3, [0.001], 4) as s:
        assert s.launches_per_iteration() == 2
        with pytest.raises(j.J2PError, match="single-launch"):
            s.debug_option(j.J2P_OPT_FUSE, 1)


def test_where_the_switch_applies(exp_lib):
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    big = synth.make_planes(4096, 4096, "444", 10, seed=3, y_only=True)
    with j.Solver(big, 0.3, [0.001], 4) as s:
        assert s.launches_per_iteration() == 3
        s.debug_option(j.J2P_OPT_FUSE, 1)
        assert s.launches_per_iteration() == 1
    joint = synth.make_planes(512, 512, "420", 10, seed=3)
    with j.Solver(joint, 0.3, [0.001] * 3, 4) as s:
        assert s.launches_per_iteration() == 2
        with pytest.raises(j.J2PError, match="single-launch"):
            s.debug_option(j.J2P_OPT_FUSE, 1)


@pytest.mark.timeout(300)
def test_three_fused_solvers_share_the_chip(exp_lib, oracle):
    """configs[1]'s shape: the three components of a 1080p 4:4:4 image as three compute(1, ...) solves in flight at once
    (jpeg2png.c:147-152), each iterating with one launch per iteration — waiting gradient wavefronts of one solver next to
    the projection workgroups of another; plus logged chunks in between (they take the two-launch form)"""
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    planes = synth.make_planes(1920, 1080, "444", 10, seed=77)
    for p in planes:
        p.fdata = oracle.decode_plane(p)
    weights = [0.3, 0.0, 0.0]
    its = 40
    wants = [reference(oracle, [planes[c]], weights[c], 0.001, its)[0] for c in range(3)]
    solvers = [j.Solver([planes[c]], weights[c], [0.001], its) for c in range(3)]
    for s in solvers:
        s.debug_option(j.J2P_OPT_FUSE, 1)
    errs = []

    def work(s):
        try:
            assert s.launches_per_iteration() == 1
            s.run(17)
            s.run(3, log=True)
            s.run(20)
            s.sync()
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(s,)) for s in solvers]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for c, s in enumerate(solvers):
        assert bit_equal(s.download(0), wants[c]), f"component {c}"
        s.close()
