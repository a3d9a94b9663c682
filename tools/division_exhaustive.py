#!/usr/bin/env python3
"""Runs the exhaustive checks of the short division on the GPU (about a minute in all):
  pass 3: all 2^23 denominator mantissas (reciprocal by IEEE division) x all 2^23 numerator mantissas — must be clean;
  pass 1 / pass 2: the reciprocal phase A could refine from its v_rsq_f32 seed, and the quotients made with it —
  expected to fail exactly on norms with an all-ones mantissa (which is why phase A keeps the long division).
usage: python tools/division_exhaustive.py [3|12|all]"""
import json
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import jpeg2png_amd as j        # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"


def sweep(which, total, size=1 << 16):
    bad, offenders, t0 = 0, [], time.time()
    for first in range(0, total, size):
        b, off = j.division_exhaustive(which, first, min(size, total - first))
        bad += b
        offenders += off
    return bad, offenders[:8], round(time.time() - t0, 1)


if what in ("3", "all"):
    bad, off, secs = sweep(3, 1 << 23)
    print(json.dumps({"pass": 3, "denominators": 1 << 23, "numerators_each": 1 << 23, "reciprocal": "1.f / d (IEEE)",
                      "quotient_mismatches": bad, "first": off, "seconds": secs}), flush=True)
if what in ("12", "all"):
    t0 = time.time()
    bad, off = j.division_exhaustive(1)
    print(json.dumps({"pass": 1, "radicands": "every float in [2^-100, 2^127)", "reciprocal": "v_rsq_f32 seed + two Newton steps",
                      "reciprocal_mismatches": bad, "first": off, "seconds": round(time.time() - t0, 2)}), flush=True)
    bad, off, secs = sweep(2, 1 << 24)
    print(json.dumps({"pass": 2, "radicands": 1 << 24, "numerators_each": 1 << 23, "quotient_mismatches": bad, "first": off,
                      "seconds": secs}), flush=True)
