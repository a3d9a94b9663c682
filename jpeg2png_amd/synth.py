"""Deterministic synthetic "JPEG coefficient" inputs for tests and bench.py.

No libjpeg needed: an image is synthesised (smooth background + hard-edged discs +
sigma=2 noise: the cartoon-like content jpeg2png targets, reference README.md:43-46),
converted to level-shifted YCbCr, chroma box-averaged for 4:2:0, padded to whole 8x8
blocks by edge replication, transformed with an orthonormal 8x8 DCT and quantised with
the IJG Annex-K tables scaled by the libjpeg quality rule.  The result
is exactly what read_jpeg() would hand over (jpeg.c:49-77): block-major int16
coefficients + a uint16 quantisation table per component.

Rows can be generated band by band (`rows=(y0, y1)`) with identical content, so
every rank of a row-tiled run synthesises only its own band.
"""
from dataclasses import dataclass
import numpy as np

# ITU-T T.81 Annex K, tables K.1 / K.2 (natural order)
_LUMA = np.array([
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55,
    14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
    18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99])
_CHROMA = np.array([
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99])


def quant_table(kind, quality):
    """IJG quality scaling (libjpeg jcparam.c rule; not part of the reference)."""
    base = _LUMA if kind == "luma" else _CHROMA
    scale = 5000 // quality if quality < 50 else 200 - 2 * quality
    return np.clip((base * scale + 50) // 100, 1, 255).astype(np.uint16)


@dataclass
class Plane:
    """One colour component as compute() receives it (struct coef, jpeg2png.h:7-20)."""
    w: int
    h: int
    w_samp: int
    h_samp: int
    data: np.ndarray          # int16, block-major [h/8][w/8][64] flattened
    quant_table: np.ndarray   # uint16[64]
    fdata: np.ndarray = None  # float32 [h, w] decoded plane (jpeg.c:83-92 + unbox), filled by a decoder


def _dct_matrix():
    k = np.arange(8)[:, None]
    n = np.arange(8)[None, :]
    m = np.cos((2 * n + 1) * k * np.pi / 16) * 0.5
    m[0, :] *= 1 / np.sqrt(2)
    return m


_M = _dct_matrix()


def _disc_list(W, H, seed):
    rng = np.random.default_rng([seed, 0xD15C])
    n = max(W, H) // 100 + 3
    cx = rng.uniform(0, W, n)
    cy = rng.uniform(0, H, n)
    r = rng.uniform(max(W, H) / 60 + 2, max(W, H) / 12 + 4, n)
    col = rng.uniform(20, 235, (n, 3))
    return cx, cy, r, col


def synth_rgb(W, H, seed, rows=None):
    """float32 [rows, W, 3] in 0..255; deterministic in (W, H, seed) and independent of the row split."""
    y0, y1 = (0, H) if rows is None else rows
    yy = np.arange(y0, y1, dtype=np.float64)[:, None]
    xx = np.arange(W, dtype=np.float64)[None, :]
    img = np.empty((y1 - y0, W, 3), dtype=np.float32)
    for c, (fx, fy, ph) in enumerate([(1.0, 0.7, 0.0), (0.6, 1.1, 1.3), (0.9, 0.5, 2.1)]):
        bg = 128 + 60 * np.sin(2 * np.pi * fx * xx / W + ph) * np.cos(2 * np.pi * fy * yy / H) \
            + 30 * (xx / W - 0.5) + 20 * (yy / H - 0.5)
        img[:, :, c] = bg
    cx, cy, r, col = _disc_list(W, H, seed)
    for i in range(len(cx)):
        ya, yb = int(max(y0, np.floor(cy[i] - r[i]))), int(min(y1, np.ceil(cy[i] + r[i]) + 1))
        if ya >= yb:
            continue
        xa, xb = int(max(0, np.floor(cx[i] - r[i]))), int(min(W, np.ceil(cx[i] + r[i]) + 1))
        if xa >= xb:
            continue
        sub_y = np.arange(ya, yb)[:, None]
        sub_x = np.arange(xa, xb)[None, :]
        mask = (sub_x - cx[i]) ** 2 + (sub_y - cy[i]) ** 2 <= r[i] ** 2
        img[ya - y0:yb - y0, xa:xb][mask] = col[i]
    # noise per 64-row strip so that any band split reproduces the same image
    for s0 in range(y0 - y0 % 64, y1, 64):
        rng = np.random.default_rng([seed, 0x0153, s0 // 64])
        strip = rng.normal(0.0, 2.0, (64, W, 3)).astype(np.float32)
        a, b = max(s0, y0), min(s0 + 64, y1)
        img[a - y0:b - y0] += strip[a - s0:b - s0]
    return np.clip(img, 0, 255)


def _pad8(p):
    h, w = p.shape
    return np.pad(p, ((0, (-h) % 8), (0, (-w) % 8)), mode="edge")


def encode_plane(pix, q):
    """pix: float [h, w] (multiples of 8, level shifted) -> int16 block-major coefficients."""
    h, w = pix.shape
    out = np.empty((h // 8, w // 8, 8, 8), dtype=np.int16)
    qd = q.reshape(8, 8).astype(np.float64)
    step = max(1, (1 << 22) // (w * 8))          # bound the float64 temporaries for very large planes
    for b0 in range(0, h // 8, step):
        b1 = min(h // 8, b0 + step)
        b = pix[b0 * 8:b1 * 8].reshape(b1 - b0, 8, w // 8, 8).transpose(0, 2, 1, 3).astype(np.float64)
        c = np.matmul(np.matmul(_M, b), _M.T)
        out[b0:b1] = np.clip(np.rint(c / qd), -32768, 32767).astype(np.int16)
    return out.reshape(-1)


def make_planes(W, H, subsampling="444", quality=10, seed=1234, y_only=False, rows=None):
    """Synthesise the components of a W x H image (or of the row band `rows`, which must
    be aligned to 16 rows).  Returns a list of Plane (1 for y_only, else 3: Y, Cb, Cr).
    For a band the planes describe only the band's rows (h = band rows / h_samp)."""
    rgb = synth_rgb(W, H, seed, rows)
    r, g, b = rgb[:, :, 0].astype(np.float64), rgb[:, :, 1].astype(np.float64), rgb[:, :, 2].astype(np.float64)
    del rgb
    y = 0.299 * r + 0.587 * g + 0.114 * b - 128.0
    planes = []
    qy = quant_table("luma", quality)
    yp = _pad8(y)
    planes.append(Plane(yp.shape[1], yp.shape[0], 1, 1, encode_plane(yp, qy), qy))
    if y_only:
        return planes
    cb = -0.168736 * r - 0.331264 * g + 0.5 * b
    cr = 0.5 * r - 0.418688 * g - 0.081312 * b
    qc = quant_table("chroma", quality)
    s = {"444": (1, 1), "420": (2, 2), "422": (2, 1), "440": (1, 2), "411": (4, 1), "410": (4, 2)}[subsampling]
    for p in (cb, cr):
        if s != (1, 1):
            hh, ww = p.shape
            p = np.pad(p, ((0, (-hh) % s[1]), (0, (-ww) % s[0])), mode="edge")
            p = p.reshape(p.shape[0] // s[1], s[1], p.shape[1] // s[0], s[0]).mean(axis=(1, 3))
        pp = _pad8(p)
        planes.append(Plane(pp.shape[1], pp.shape[0], s[0], s[1], encode_plane(pp, qc), qc))
    return planes


def make_y_plane_banded(W, H, quality, seed, band_rows=2048, workers=8):
    """A large Y-only plane synthesised band by band in worker processes (identical to make_planes(..., y_only=True)
    of the whole image: the content does not depend on the row split) — bounds the float64 temporaries and uses the
    host's cores.  H must be a multiple of 8; band_rows a multiple of 64.  The workers are fresh interpreters
    (`python -m jpeg2png_amd.synth ...`), not forks: the caller may hold an initialised HIP runtime."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tmp = tempfile.mkdtemp(prefix="j2p_synth_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    jobs = [(r0, min(H, r0 + band_rows)) for r0 in range(0, H, band_rows)]
    parts = [None] * len(jobs)
    try:
        running = {}
        nxt = 0
        while nxt < len(jobs) or running:
            while nxt < len(jobs) and len(running) < workers:
                r0, r1 = jobs[nxt]
                out = os.path.join(tmp, f"{nxt}.npy")
                running[nxt] = (subprocess.Popen([sys.executable, "-W", "ignore::RuntimeWarning", "-m", "jpeg2png_amd.synth", str(W), str(H), str(quality), str(seed),
                                                  str(r0), str(r1), out], cwd=root), out)
                nxt += 1
            for k in list(running):
                proc, out = running[k]
                if proc.poll() is None:
                    continue
                if proc.returncode != 0:
                    raise RuntimeError(f"synth worker for band {k} failed ({proc.returncode})")
                parts[k] = np.load(out)
                os.unlink(out)
                del running[k]
            if running:
                next(iter(running.values()))[0].wait()
    finally:
        for proc, _ in running.values():
            proc.kill()
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return Plane(W, H, 1, 1, np.concatenate(parts), quant_table("luma", quality))


if __name__ == "__main__":
    import sys
    _W, _H, _q, _seed, _r0, _r1 = (int(v) for v in sys.argv[1:7])
    np.save(sys.argv[7], make_planes(_W, _H, "444", _q, seed=_seed, y_only=True, rows=(_r0, _r1))[0].data)
