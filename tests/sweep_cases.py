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
