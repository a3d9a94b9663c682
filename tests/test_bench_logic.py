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
