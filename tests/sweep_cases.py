"""Randomised solver configurations shared by tools/sweep_vs_ref.py and the GPU parity tests: shapes from 1x1
to ~1400x1100, every sampling the synthesiser knows, qualities 5..98, TV-only / TGV weights, per-channel
pweights including 0, optional flat grey areas (all-zero coefficient blocks), optional logging."""
from dataclasses import dataclass

import numpy as np


@dataclass
class SweepCase:
    index: int
    W: int
    H: int
    sub: str
    quality: int
    y_only: bool
    iterations: int
    weight: float
    plane_seed: int
    flat: bool
    pweights: list
    log: bool

    def describe(self):
        return (f"{self.index:3d} {self.W:4d}x{self.H:<4d} {self.sub} q{self.quality:<2d} {'Y' if self.y_only else 'YCC'} "
                f"its {self.iterations:2d} w {self.weight} pw {self.pweights} log {int(self.log)} flat {int(self.flat)}")

    def planes(self):
        """synthetic planes of the case (fdata not decoded yet)"""
        from jpeg2png_amd import synth
        planes = synth.make_planes(self.W, self.H, self.sub, self.quality, seed=self.plane_seed, y_only=self.y_only)
        if self.flat:
            for p in planes[1:] if len(planes) > 1 else planes:
                d = p.data.reshape(p.h // 8, p.w // 8, 64)
                d[: max(1, d.shape[0] // 2), : max(1, d.shape[1] // 2)] = 0
        return planes


def cases(seed, n):
    """the first n cases of stream `seed` (case i is the same whatever n is)"""
    rng = np.random.default_rng(seed)
    for i in range(n):
        sub = str(rng.choice(["444", "420", "422", "440", "411", "410"]))
        W = int(rng.integers(17, 1400))
        H = int(rng.integers(17, 1100))
        if rng.random() < 0.3:
            W, H = int(rng.integers(1, 80)), int(rng.integers(1, 80))
        q = int(rng.choice([5, 10, 30, 50, 75, 90, 98]))
        y_only = bool(rng.random() < 0.25)
        its = int(rng.integers(1, 40))
        weight = float(rng.choice([0.0, 0.1, 0.3, 1.0]))
        pseed = int(rng.integers(1 << 30))
        flat = bool(rng.random() < 0.3)
        pws = [float(rng.choice([0.0, 0.001, 0.01])) for _ in range(1 if y_only else 3)]
        log = bool(rng.random() < 0.3)
        yield SweepCase(i, W, H, sub, q, y_only, its, weight, pseed, flat, pws, log)


def case(seed, index):
    for c in cases(seed, index + 1):
        pass
    return c


def cases_wide(seed, n):
    """a second stream with wider ranges: sizes to ~3000, continuous weights, qualities 1..100, up to 120
    iterations, quantisation tables scaled into the 16-bit range (below 32768), sparse data"""
    rng = np.random.default_rng([seed, 0xA11CE])
    for i in range(n):
        sub = str(rng.choice(["444", "420", "422", "440", "411", "410"]))
        big = rng.random() < 0.15
        W = int(rng.integers(8, 3000 if big else 700))
        H = int(rng.integers(8, 2200 if big else 500))
        q = int(rng.integers(1, 101))
        y_only = bool(rng.random() < 0.3)
        its = int(rng.integers(1, 30 if big else 120))
        weight = float(rng.choice([0.0, float(rng.uniform(0, 2))]))
        pseed = int(rng.integers(1 << 30))
        flat = bool(rng.random() < 0.3)
        pws = [float(rng.choice([0.0, float(10 ** rng.uniform(-5, -1))])) for _ in range(1 if y_only else 3)]
        log = bool(rng.random() < 0.3)
        c = SweepCase(i, W, H, sub, q, y_only, its, weight, pseed, flat, pws, log)
        c.qscale = int(rng.choice([1, 1, 1, 7, 100]))          # 16-bit-precision tables
        c.sparsify = bool(rng.random() < 0.2)                  # keep only DC and a few AC coefficients
        yield c


def planes_wide(cs):
    planes = cs.planes()
    rng = np.random.default_rng(cs.plane_seed)
    for p in planes:
        if getattr(cs, "qscale", 1) != 1:
            p.quant_table = np.minimum(p.quant_table.astype(np.int64) * cs.qscale, 30000).astype(np.uint16)
        if getattr(cs, "sparsify", False):
            d = p.data.reshape(-1, 64)
            keep = rng.random(64) < 0.15
            keep[0] = True
            d[:, ~keep] = 0
    return planes
