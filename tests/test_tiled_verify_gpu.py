"""j2p_tiled_create checks and picks its exchange on the GPUs it is given (jpeg2png_amd/csrc/j2p_tiled.hip, pick_plan):
a scratch canvas cut from the job's own first rows is solved whole by one plain solver — no exchange, the truth — and as
bands through every candidate exchange; a candidate that differs in one bit is demoted, the fastest of the rest is the
plan for that device list.  On bands that each have a GPU of their own this runs by itself; on this pool's one-GPU boxes
J2P_TILED_VERIFY=1 runs it with the bands sharing the GPU (the rehearsal), and a fault-injection build
(-DJ2P_EXP_DROP_HALO_PUSH: k_project does not push its edge rows) shows that a broken exchange IS found and replaced.
(Loop being tiled: compute.c:427-453.)"""
import copy
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import band_devices, bit_equal, make_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys, copy
sys.path.insert(0, %(root)r)
import numpy as np
import jpeg2png_amd as j
from jpeg2png_amd import synth
from oracle import bindings as ob
planes = synth.make_planes(%(W)d, %(H)d, %(sub)r, 10, seed=%(seed)d, y_only=%(y_only)r)
for p in planes:
    p.fdata = ob.decode_plane(p)
pws = [0.001] * len(planes)
its = 9
whole = copy.deepcopy(planes)
j.compute(whole, 0.3, pws, its)
out = {}
with j.TiledSolver(planes, 0.3, pws, its, devices=%(devices)r) as t:
    out["exchange"] = t.exchange()
    t.run(its)
    out["equal"] = [bool(np.array_equal(t.download(c).view(np.uint32), whole[c].fdata.view(np.uint32))) for c in range(len(planes))]
with j.TiledSolver(planes, 0.3, pws, its, devices=%(devices)r) as t:      # the plan is cached per device list: no second verification
    out["exchange_again"] = t.exchange()
print("RESULT " + json.dumps(out))
"""


def run_child(env_extra, devices, W=264, H=410, sub="420", seed=83, y_only=False):
    env = dict(os.environ, **env_extra)
    for k in ("J2P_TILED_EXCHANGE", "J2P_TILED_WAIT"):
        env.pop(k, None)
    code = CHILD % {"root": ROOT, "W": W, "H": H, "sub": sub, "seed": seed, "y_only": y_only, "devices": devices}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads(line[-1][7:]), r.stderr


@pytest.mark.timeout(600)
def test_the_picker_verifies_every_candidate_and_keeps_a_right_one(lib):
    """release library, two bands sharing this GPU, J2P_TILED_VERIFY=1: every candidate reproduces the one-GPU solve (none is
    demoted), the fastest becomes the plan, the planes of the real job equal the whole-canvas solve; on a box with several
    GPUs the same runs unasked with a GPU per band"""
    devices = band_devices(2)
    out, err = run_child({"J2P_TILED_VERIFY": "2"}, devices)        # (2: verify — also on a shared GPU — and say what was measured)
    assert "DEMOTED" not in err, err
    assert "verified per scratch iteration" in err and "'direct'" in err and "'copy'" in err, err
    assert out["exchange"] in ("direct, wait counter", "direct, wait collector", "direct", "copy", "rccl")
    assert out["exchange_again"] == out["exchange"]
    assert err.count("verified per scratch iteration") == 1, "the second create on the same devices verified again"
    assert all(out["equal"]), out
    print("picked:", out["exchange"], "|", [ln for ln in err.splitlines() if "verified per scratch" in ln][-1])


@pytest.mark.timeout(900)
def test_a_broken_exchange_is_found_and_demoted(lib):
    """fault injection: a library whose k_project does NOT push its edge rows into the neighbours' halo rows
    (-DJ2P_EXP_DROP_HALO_PUSH, built here by tools/build_variant.py).  Both `direct` candidates must be demoted — their
    scratch canvases differ from the one-GPU solve — `copy`, which pulls the rows with a kernel of its own, must be what
    the job then runs on, and the job's planes must be right."""
    lib_path = _variant("drop_halo_push", "-DJ2P_EXP_DROP_HALO_PUSH")
    devices = band_devices(2)
    own_gpus = len(set(devices)) == len(devices)
    env = {"J2P_LIBRARY": lib_path, "J2P_TILED_VERIFY": "2"}
    out, err = run_child(env, devices)
    assert "exchange 'direct' DEMOTED" in err, err
    assert "exchange 'direct, wait counter' DEMOTED" in err or "wait counter' not available" in err, err
    assert out["exchange"] == "copy" and out["exchange_again"] == "copy"
    assert all(out["equal"]), out
    # ... and WITHOUT the verification the broken exchange goes unnoticed by the library: that is what it is for
    if not own_gpus:
        out2, _ = run_child({"J2P_LIBRARY": lib_path, "J2P_TILED_VERIFY": "0"}, devices)
        assert out2["exchange"] == "direct" and not all(out2["equal"])


def _variant(name, flag):
    lib_path = os.path.join(ROOT, "ab", f"libj2p_{name}.so")
    sources = [os.path.join(ROOT, "jpeg2png_amd", "csrc", f) for f in ("j2p_kernels.hip.h", "j2p_solver.hip", "j2p_tiled.hip", "j2p_batch.hip", "compute_host.c")]
    if not os.path.exists(lib_path) or any(os.path.getmtime(f) > os.path.getmtime(lib_path) for f in sources):
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), name, flag], check=True, cwd=ROOT, timeout=600)
    return lib_path


@pytest.mark.timeout(900)
def test_an_exchange_that_never_finishes_costs_its_candidacy_not_the_process(lib):
    """fault injection no. 2: a library whose k_gradient never counts up the values the `counter` form waits for
    (-DJ2P_EXP_DROP_COUNTS) — on hardware this code has not met, THAT is what a wrong assumption about stream memory
    operations would look like: the candidate does not fail, it never finishes.  The verification runs every candidate
    against a deadline: `direct, wait counter` must be reported as not finishing, its stuck streams released and torn
    down, another exchange picked, and the job's planes right."""
    devices = band_devices(2)
    out, err = run_child({"J2P_LIBRARY": _variant("drop_counts", "-DJ2P_EXP_DROP_COUNTS"), "J2P_TILED_VERIFY": "2"}, devices)
    assert "exchange 'direct, wait counter' not available" in err and "did not finish" in err, err
    assert out["exchange"] in ("direct", "copy") and out["exchange_again"] == out["exchange"]
    assert all(out["equal"]), out


@pytest.mark.timeout(120)
@pytest.mark.parametrize("wait", ["all", "counter"])
def test_a_band_that_fails_mid_run_does_not_hang_the_others(lib, wait, monkeypatch):
    """teardown of a failed row-tiled run: the last band stops halfway through its iterations (injected) while the other
    band's launches — and, in the value form, its hipStreamWaitValue64 on counts that will never come — are queued.  run()
    reports the band's error, sync() says the solver is unusable instead of blocking, destroy returns."""
    import ctypes
    import jpeg2png_amd as j
    planes = make_case(200, 330, "420", 10, seed=84)
    monkeypatch.setenv("J2P_TILED_EXCHANGE", "direct")
    monkeypatch.setenv("J2P_TILED_WAIT", wait)
    lib.j2p_debug_fail_run_after.argtypes = [ctypes.c_int]
    lib.j2p_debug_fail_run_after.restype = None
    t = j.TiledSolver(planes, 0.3, [0.001] * 3, 12, devices=band_devices(2))
    try:
        lib.j2p_debug_fail_run_after(-1)
        with pytest.raises(j.J2PError, match="injected band failure"):
            t.run(12)
        with pytest.raises(j.J2PError, match="unusable"):
            t.sync()
        with pytest.raises(j.J2PError, match="unusable"):
            t.run(1)
    finally:
        lib.j2p_debug_fail_run_after(0)
        t.close()
    # the library is fine afterwards
    want = copy.deepcopy(planes)
    j.compute(want, 0.3, [0.001] * 3, 5)
    with j.TiledSolver(planes, 0.3, [0.001] * 3, 5, devices=band_devices(2)) as t2:
        t2.run(5)
        for c in range(3):
            assert bit_equal(t2.download(c), want[c].fdata)
