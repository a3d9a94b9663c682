import sys
import numpy as np
sys.path.insert(0, ".")
import jpeg2png_amd as j
from oracle import bindings as oracle
rng = np.random.default_rng(3)
n = 4096
blocks = np.zeros((n, 64), dtype=np.float32)
# tiny entries: a few per block, magnitudes from subnormal to just-normal, random signs
for b in range(n):
    k = rng.integers(1, 6)
    idx = rng.integers(0, 64, k)
    mag = np.exp(rng.uniform(np.log(1e-45), np.log(1e-36), k))
    blocks[b, idx] = (mag * rng.choice([-1, 1], k)).astype(np.float32)
for inverse in (False, True):
    got = j.dct8x8_blocks(blocks, inverse=inverse)
    want = oracle.dct_blocks(blocks, inverse=inverse)
    dm = got.view(np.uint32) != want.view(np.uint32)
    print("inverse" if inverse else "forward", "blocks differing:", int(dm.any(axis=1).sum()), "of", n, "elements:", int(dm.sum()))
    if dm.any():
        b = int(np.nonzero(dm.any(axis=1))[0][0])
        e = np.nonzero(dm[b])[0]
        print(" block", b, "input nonzeros", {int(i): float(blocks[b, i]) for i in np.nonzero(blocks[b])[0]})
        print(" elems", e[:8], "gpu", [hex(int(x)) for x in got.view(np.uint32)[b, e[:8]]], "cpu", [hex(int(x)) for x in want.view(np.uint32)[b, e[:8]]])
