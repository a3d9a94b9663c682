#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the UNMODIFIED reference solver.

Runs only where /root/reference exists (oracle/Makefile builds oracle/_ref from it).
Each fixture stores the inputs (block-major int16 coefficients, quant tables, decoded
planes produced by the reference's own idct8x8s via decode) and the outputs of the
reference's compute(): canvas planes plus the CSV log columns.  The committed files pin
the CPU oracle (and through it the HIP path) on machines without the reference sources.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from jpeg2png_amd import synth  # noqa: E402
from oracle import bindings as ob  # noqa: E402

# name, W, H, subsampling, quality, y_only, weight, pweight(s), iterations, seed
CASES = [
    ("y_48x40_tgv", 48, 40, "444", 10, True, 0.3, [0.001], 12, 101),
    ("y_40x24_tvonly", 40, 24, "444", 10, True, 0.0, [0.001], 8, 102),
    ("y_32x32_noprob", 32, 32, "444", 25, True, 0.3, [0.0], 8, 103),
    ("rgb420_64x48", 64, 48, "420", 10, False, 0.3, [0.001] * 3, 10, 104),
    ("rgb420_padded_40x20", 40, 20, "420", 10, False, 0.3, [0.001] * 3, 10, 105),
    ("rgb444_48x32_q50", 48, 32, "444", 50, False, 0.3, [0.001, 0.002, 0.0], 6, 106),
    ("rgb422_48x32", 48, 32, "422", 10, False, 0.3, [0.001] * 3, 6, 107),
]


def main():
    ob.build(ref=True)
    assert ob.have_ref(), "needs /root/reference"
    rng = np.random.default_rng(7)
    blocks = np.concatenate([rng.normal(0, 60, (48, 64)), rng.integers(-512, 512, (16, 64))]).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "dct_blocks.npz"), blocks=blocks,
                        fdct=ob.dct_blocks(blocks, False, "ref"), idct=ob.dct_blocks(blocks, True, "ref"))
    for name, W, H, sub, q, y_only, weight, pw, its, seed in CASES:
        planes = synth.make_planes(W, H, sub, q, seed=seed, y_only=y_only)
        for p in planes:
            p.fdata = ob.decode_plane(p)
        outs, rows, _ = ob.ref_compute(planes, weight, pw, its, log=True)
        d = {"weight": np.float32(weight), "pweight": np.array(pw, np.float32), "iterations": np.int32(its),
             "log": rows}
        for c, p in enumerate(planes):
            d[f"geom{c}"] = np.array([p.w, p.h, p.w_samp, p.h_samp], np.int32)
            d[f"data{c}"] = p.data
            d[f"quant{c}"] = p.quant_table
            d[f"fdata{c}"] = p.fdata
            d[f"out{c}"] = outs[c]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, [o.shape for o in outs])


if __name__ == "__main__":
    main()
