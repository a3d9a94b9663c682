"""bench.py --gpus N: which leg of the row-tiled run becomes `value` (no GPU needed: the rule itself)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def legs(**kw):
    return {k: {"elapsed": e, "digest": d} for k, (e, d) in kw.items()}


def test_fastest_verified_leg_wins_and_wrong_planes_never_do():
    import bench
    t = legs(c=(1.0, "good"), c_copy=(1.3, "good"), c_counter=(0.8, "BAD"), rccl=(0.7, None))
    assert bench.pick_leg(t, "good") == "c"
    assert t["c"]["verified"] is True and t["c_counter"]["verified"] is False and t["rccl"]["verified"] is None


def test_without_a_verified_leg_the_unverified_ones_count_before_the_wrong_ones():
    import bench
    t = legs(c=(1.0, "BAD"), rccl=(2.0, None))
    assert bench.pick_leg(t, "good") == "rccl"
    t = legs(c=(1.0, "BAD"), c_copy=(2.0, "WORSE"))
    assert bench.pick_leg(t, "good") == "c"               # nothing else exists: reported with verified == False


def test_without_a_truth_the_fastest_leg_wins():
    import bench
    t = legs(c=(1.0, "x"), c_copy=(0.9, "y"))
    assert bench.pick_leg(t, None) == "c_copy"
    assert t["c"]["verified"] is None


# ---------------------------------------------------------------------------------------------------------------
# tests/golden/bench_digests.json: what bench.py's `parity` object compares the timed solver's plane with
# ---------------------------------------------------------------------------------------------------------------
def _digests():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "bench_digests.json")) as f:
        return json.load(f)


def test_bench_digest_file_describes_the_workloads_bench_py_times():
    import bench
    d = _digests()
    for name, (W, H, its, seed) in {"configs[2]": (4096, 4096, 500, 1234 + 3), "configs[3] N=1": (16384, 2048, 100, 1234 + 4)}.items():
        e = d[name]
        assert (e["W"], e["H"], e["iterations"], e["seed"], e["quality"]) == (W, H, its, seed, 10)
        assert (e["weight"], e["pweight"]) == (bench.WEIGHT, bench.PWEIGHT)
        assert len(e["digest"]) == 32
        assert bench.reference_digest(W, H, its, seed) == (name, e["digest"])
    assert bench.reference_digest(4096, 4096, 499, 1234 + 3) == (None, None)
    ok = bench.parity_object(d["configs[2]"]["digest"], 4096, 4096, 500, 1234 + 3)
    assert ok["bit_identical"] is True and ok["entry"] == "configs[2]"
    assert bench.parity_object("0" * 32, 4096, 4096, 500, 1234 + 3)["bit_identical"] is False
    assert bench.parity_object("0" * 32, 1024, 1024, 5, 1)["bit_identical"] is None


def test_bench_digest_recipe_reproduces_from_the_compiled_reference():
    """the generator's small entry — rows 0..256 of the configs[2] image, 10 iterations — recomputed here with oracle/_ref:
    same synthesis, same decode, same reference, same hash function as the entries bench.py relies on"""
    import hashlib
    import numpy as np
    import pytest
    from jpeg2png_amd import synth
    from oracle import bindings as ob
    if not ob.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    e = _digests()["recipe"]
    plane = synth.make_planes(e["W"], e["image_H"], "444", e["quality"], seed=e["seed"], y_only=True, rows=tuple(e["rows_of_the_image"]))[0]
    plane.fdata = ob.decode_plane(plane)
    outs, _, _ = ob.ref_compute([plane], e["weight"], [e["pweight"]], e["iterations"])
    assert hashlib.blake2b(np.ascontiguousarray(outs[0]), digest_size=16).hexdigest() == e["digest"]


# ---------------------------------------------------------------------------------------------------------------
# the fields of the bench line that point INTO profiles/: they must name files that exist and say what they are
# ---------------------------------------------------------------------------------------------------------------
def test_counter_traffic_comes_from_the_committed_profiles_by_shape():
    import bench
    for shape in ((4096, 4096), (8192, 8192), (16384, 4096), (16384, 8192)):
        traffic, src = bench.pmc_traffic(*shape)
        px = shape[0] * shape[1]
        assert src and os.path.exists(os.path.join(ROOT, src.split(" ")[0])), shape
        # real bytes per iteration sit a little above the 38 algorithmic ones (halo rows), never below, never 1.2 x above
        assert 38 * px <= traffic <= 1.2 * 38 * px, (shape, traffic / px)
    assert bench.pmc_traffic(1000, 1000) == (None, None)            # a shape nobody profiled: null, not a guess


def test_band_traffic_is_for_the_band_shape_only():
    import bench
    traffic, src = bench.band_pmc_traffic(16384, 2048)
    assert src and src.startswith("profiles/r06_pmc_band.json") and os.path.exists(os.path.join(ROOT, "profiles", "r06_pmc_band.json"))
    assert 38 * 16384 * 2048 <= traffic <= 1.2 * 38 * 16384 * 2048
    assert bench.band_pmc_traffic(4096, 512) == (None, None)        # the debug size of the rehearsal: no counters on file


def test_roofline_object_says_which_fraction_is_which():
    import bench
    per_kernel = bench.per_kernel_roofline(4096 * 4096, 4096 * 4096, 0.056, 0.061)
    r = bench.roofline_object(143000.0, 1, 500, 0.0585 * 25, 25, 4096 * 4096, per_kernel, 100, 16, 676000000, "profiles/x.json")
    assert "kernel_frac" not in r                                   # (round 5's name: read as the whole iteration's by some)
    assert r["kernel"] == "k_gradient" and r["event_kernel_frac"] == per_kernel["k_gradient"]["frac"]
    assert abs(r["frac"] - 38 * 143000.0e6 / 8e12) < 1e-3 and r["traffic"] == 676000000 and r["traffic_source"] == "profiles/x.json"
    prof, src = bench.rocprof_kernel_us()
    assert prof and src and os.path.exists(os.path.join(ROOT, src)) and 30 < prof["k_gradient"] < 80 and 40 < prof["k_project"] < 80
