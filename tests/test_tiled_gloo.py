"""world_size-2 CPU test (gloo) of the row-tiling driver jpeg2png_amd/tiled.py: the halo
exchange and the all-gather of norm partials are exercised with a CPU band engine that has
the same data dependencies as the HIP solver (gradient reaches 2 rows, norm over the
GLOBAL partial array in a fixed order, block-local update).  The 2-rank result must equal
the 1-rank result bit for bit — the property the GPU path relies on for GPU-count
invariance.  (The HIP band kernels themselves are checked on one GPU by
test_parity_gpu.py::test_band_split_matches_whole.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HALO, TILE = 2, 16


class CpuBandEngine:
    """Same interface as tiled.HipBandEngine, arithmetic in torch on the CPU.
    gradient(y) = second vertical + horizontal differences (reach 2 rows), partial sums of
    g*g per 16-row tile row; project: x_new = y - step*g/norm with norm from the fixed-order
    tree over ALL tile rows; FISTA point y = x + f*(x - xprev) formed locally incl. halos."""

    def __init__(self, full, band, iterations):
        H, W = full.shape
        self.r0, self.r1 = band
        self.H, self.W, self.nch = H, W, 1
        rows = self.r1 - self.r0
        self.rows = rows
        self.buf = [torch.zeros(rows + 2 * HALO, W, dtype=torch.float32) for _ in range(2)]
        self.buf[0][HALO:HALO + rows] = full[self.r0:self.r1]
        self.buf[1][HALO:HALO + rows] = full[self.r0:self.r1]
        self.cur = 0
        self.local_tile_rows = rows // TILE
        self.global_tile_rows = H // TILE
        self.first_tile_row = self.r0 // TILE
        self.partials_local = torch.zeros(self.local_tile_rows, dtype=torch.float64)
        self.partials_all = torch.zeros(self.global_tile_rows, dtype=torch.float64)
        self.t = 1.0
        self.step = 0.5 / np.sqrt(1 + iterations)
        self.g = torch.zeros(rows, W, dtype=torch.float32)

    def stream_context(self):
        import contextlib
        return contextlib.nullcontext()

    def halo(self):
        b = self.buf[self.cur]
        r = self.rows
        return {"send_top": [b[HALO:2 * HALO].reshape(-1)], "recv_top": [b[0:HALO].reshape(-1)],
                "send_bottom": [b[r:r + HALO].reshape(-1)], "recv_bottom": [b[r + HALO:r + 2 * HALO].reshape(-1)]}

    def commit_initial_halo(self):
        self.buf[self.cur ^ 1][:HALO] = self.buf[self.cur][:HALO]
        self.buf[self.cur ^ 1][-HALO:] = self.buf[self.cur][-HALO:]

    def phase_gradient(self):
        tn = (1 + np.sqrt(1 + 4 * self.t * self.t)) / 2
        self.factor = np.float32((self.t - 1) / tn)
        self.t = tn
        x, xp = self.buf[self.cur], self.buf[self.cur ^ 1]
        y = x + self.factor * (x - xp)
        # rows outside the image behave as zeros; top/bottom halos of the edge bands stay zero
        yy = y
        g = torch.zeros(self.rows, self.W)
        c = yy[HALO:HALO + self.rows]
        g += 6 * c - 4 * (yy[HALO - 1:HALO - 1 + self.rows] + yy[HALO + 1:HALO + 1 + self.rows])
        g += yy[HALO - 2:HALO - 2 + self.rows] + yy[HALO + 2:HALO + 2 + self.rows]
        g[:, 1:] += c[:, 1:] - c[:, :-1]
        self.g = g
        self.y = y
        sq = (g.double() ** 2).reshape(self.local_tile_rows, TILE * self.W)
        self.partials_local.copy_(sq.sum(dim=1))

    def phase_project(self):
        v = self.partials_all.clone()
        n = 1
        while n < v.numel():
            n *= 2
        v = torch.cat([v, torch.zeros(n - v.numel(), dtype=torch.float64)])
        while v.numel() > 1:
            h = v.numel() // 2
            v = v[:h] + v[h:]
        norm = torch.sqrt(v[0].float())
        new = self.y[HALO:HALO + self.rows] - np.float32(self.step) * (self.g / norm)
        self.buf[self.cur ^ 1][HALO:HALO + self.rows] = new
        self.cur ^= 1

    def result(self):
        return self.buf[self.cur][HALO:HALO + self.rows].clone()

    # ---- the two-part phases of the overlap schedule (tiled.RowTiledSolver(overlap=True)) ----
    # Same arithmetic as above cut at the same places as the HIP engine: the gradient's interior 16-row segments
    # read no halo row; the projection's first and last 8 rows (the rows the neighbours need) come first and
    # halo() then refers to the iterate being written.
    @property
    def can_split(self):
        return self.rows >= 3 * TILE

    def _fista(self):
        tn = (1 + np.sqrt(1 + 4 * self.t * self.t)) / 2
        self.factor = np.float32((self.t - 1) / tn)
        self.t = tn

    def _gradient_rows(self, a, b):
        x, xp = self.buf[self.cur], self.buf[self.cur ^ 1]
        lo, hi = a + HALO - 2, b + HALO + 2                       # buffer rows needed for band rows [a, b)
        y = x[lo:hi] + self.factor * (x[lo:hi] - xp[lo:hi])
        n = b - a
        c = y[2:2 + n]
        g = torch.zeros(n, self.W)
        g += 6 * c - 4 * (y[1:1 + n] + y[3:3 + n])
        g += y[0:n] + y[4:4 + n]                                   # same association as phase_gradient
        g[:, 1:] += c[:, 1:] - c[:, :-1]
        self.g[a:b] = g
        self.y[a:b] = c
        sq = (g.double() ** 2).reshape((b - a) // TILE, TILE * self.W)
        self.partials_local[a // TILE:b // TILE] = sq.sum(dim=1)

    def gradient_interior(self):
        self._fista()
        self.y = torch.zeros(self.rows, self.W)
        self.g = torch.zeros(self.rows, self.W)
        self._gradient_rows(TILE, self.rows - TILE)

    def gradient_edges_inline(self, halo_ready=None):
        self._gradient_rows(0, TILE)
        self._gradient_rows(self.rows - TILE, self.rows)

    def finish_gradient(self):
        pass

    def _norm(self):
        v = self.partials_all.clone()
        n = 1
        while n < v.numel():
            n *= 2
        v = torch.cat([v, torch.zeros(n - v.numel(), dtype=torch.float64)])
        while v.numel() > 1:
            h = v.numel() // 2
            v = v[:h] + v[h:]
        return torch.sqrt(v[0].float())

    def _project_rows(self, a, b):
        new = self.y[a:b] - np.float32(self.step) * (self.g[a:b] / self.norm)
        self.buf[self.cur ^ 1][HALO + a:HALO + b] = new

    def project_boundary(self):
        self.norm = self._norm()
        self._project_rows(0, 8)
        self._project_rows(self.rows - 8, self.rows)
        self.cur ^= 1               # halo() now addresses the new iterate; project_interior writes the same buffer

    def project_interior(self):
        self.cur ^= 1
        self._project_rows(8, self.rows - 8)
        self.cur ^= 1

    def project_done_event(self):
        return None

    def comm_context(self, after):
        import contextlib
        return contextlib.nullcontext()

    def comm_done_event(self):
        return None

    def wait_halo(self, event):
        pass


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, its, bands, out_dir, overlap=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jpeg2png_amd import tiled
    torch.manual_seed(0)
    full = torch.randn(H, W, dtype=torch.float32) * 20
    eng = CpuBandEngine(full, bands[rank], its)
    drv = tiled.RowTiledSolver(eng, overlap=overlap)
    assert drv.overlap == overlap
    drv.start()
    drv.iterate(its)
    np.save(os.path.join(out_dir, f"band{rank}.npy"), eng.result().numpy())
    dist.destroy_process_group()


def _run(world, H, W, its, bands, tmp_path, overlap=False):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, H, W, its, bands, str(tmp_path), overlap), nprocs=world, join=True)
    return np.concatenate([np.load(os.path.join(tmp_path, f"band{r}.npy")) for r in range(world)], axis=0)


def test_split_rows_alignment():
    sys.path.insert(0, ROOT)
    from jpeg2png_amd import tiled
    assert tiled.split_rows(16384, 8, 16) == [(i * 2048, (i + 1) * 2048) for i in range(8)]
    b = tiled.split_rows(1088, 3, 16)
    assert b[0][0] == 0 and b[-1][1] == 1088 and all(x[0] % 16 == 0 for x in b)
    assert all(b[i][1] == b[i + 1][0] for i in range(2))
    with pytest.raises(ValueError):
        tiled.split_rows(32, 4, 16)


@pytest.mark.parametrize("bands2", [[(0, 64), (64, 128)], [(0, 48), (48, 128)]], ids=["equal", "unequal"])
def test_two_ranks_equal_one_rank(tmp_path, bands2):
    H, W, its = 128, 40, 5
    one = tmp_path / "one"
    two = tmp_path / "two"
    one.mkdir()
    two.mkdir()
    a = _run(1, H, W, its, [(0, H)], one)
    b = _run(2, H, W, its, bands2, two)
    assert a.shape == b.shape == (H, W)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("bands2", [[(0, 64), (64, 128)], [(0, 48), (48, 128)]], ids=["equal", "unequal"])
def test_overlap_schedule_with_two_ranks(tmp_path, bands2):
    """RowTiledSolver(overlap=True) with world_size 2: the projection's edge rows first, the halo exchange issued
    behind them, the interior of the next gradient phase before the edge segments — against the plain schedule
    on one rank"""
    H, W, its = 128, 40, 6
    one = tmp_path / "one"
    two = tmp_path / "two"
    one.mkdir()
    two.mkdir()
    a = _run(1, H, W, its, [(0, H)], one)
    b = _run(2, H, W, its, bands2, two, overlap=True)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
