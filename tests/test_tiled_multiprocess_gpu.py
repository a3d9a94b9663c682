"""Several PROCESSES on the real kernels: world size 2 under torch.distributed (gloo), each rank a HipBandEngine — the
C-ABI band solver, k_gradient / k_project on its band — on this box's one GPU, jpeg2png_amd.tiled.RowTiledSolver
carrying the two per-iteration exchanges (all-gather of the row sums of g^2, send/recv of the 2 + 2 edge rows) staged
through host memory.  The bands put together must equal the whole-canvas solver bit for bit, CSV rows included.
(tests/test_tiled_gloo.py drives the same driver with toy arithmetic on the CPU; tests/test_tiled_c_gpu.py the real
kernels from one process; this is the third corner: real arithmetic, separate address spaces.)"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case(sub, y_only):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import make_case
    return make_case(200, 256, sub, 10, seed=57, y_only=y_only)


def _worker(rank, world, port, sub, y_only, its, bands, out_dir, log):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["J2P_TILED_TRANSPORT"] = "torch"
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import jpeg2png_amd as j
    from jpeg2png_amd import tiled
    planes = _case(sub, y_only)
    r0, r1 = bands[rank]
    local = []
    for p in planes:                        # the band's own coefficient rows only (band-local host arrays)
        c0, c1 = r0 // p.h_samp, min(p.h, (r1 + p.h_samp - 1) // p.h_samp)
        d = p.data.reshape(p.h // 8, -1)[c0 // 8:c1 // 8].reshape(-1)
        local.append(j.Plane(p.w, p.h, p.w_samp, p.h_samp, d, p.quant_table, p.fdata[c0:c1]))
    pws = [0.001] * len(planes)
    eng = tiled.HipBandEngine(local, 0.3, pws, its, (r0, r1), 0)
    drv = tiled.RowTiledSolver(eng, overlap=False, log=log)
    assert drv.host_staged and drv.direct is None
    drv.start()
    drv.iterate(its)
    for c in range(len(planes)):
        np.save(os.path.join(out_dir, f"band{rank}_c{c}.npy"), eng.download(c))
    if log and rank == 0:
        np.save(os.path.join(out_dir, "rows.npy"), drv.log_rows())
    drv.close()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("sub,y_only,bands", [("444", True, [(0, 128), (128, 256)]), ("420", False, [(0, 96), (96, 256)])],
                         ids=["y_equal_bands", "420_unequal_bands"])
def test_two_processes_on_the_real_kernels_equal_the_whole_canvas(lib, tmp_path, sub, y_only, bands):
    import copy
    import torch.multiprocessing as mp
    import jpeg2png_amd as j
    from conftest import bit_equal
    its = 7
    planes = _case(sub, y_only)
    pws = [0.001] * len(planes)
    ref = copy.deepcopy(planes)
    want_rows = j.compute(ref, 0.3, pws, its, log=True)
    mp.spawn(_worker, args=(2, _free_port(), sub, y_only, its, bands, str(tmp_path), True), nprocs=2, join=True)
    for c in range(len(planes)):
        got = np.concatenate([np.load(tmp_path / f"band{r}_c{c}.npy") for r in range(2)], axis=0)
        assert bit_equal(got, ref[c].fdata), f"channel {c}"
    rows = np.load(tmp_path / "rows.npy")
    np.testing.assert_allclose(rows, want_rows, rtol=1e-9, atol=1e-12)
