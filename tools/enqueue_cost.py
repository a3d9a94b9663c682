"""host cost of the row-tiled iteration loop: time to ENQUEUE 100 iterations (no sync) vs time to run them,
lone rank as its own neighbour (J2P_TILED_SELF_NEIGHBOURS=1), 16384x2048 band"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["J2P_TILED_SELF_NEIGHBOURS"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
import torch
import torch.distributed as dist
import jpeg2png_amd as j
from jpeg2png_amd import synth, tiled

W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rows = 2048
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
planes = synth.make_planes(W, rows, "444", 10, seed=5, y_only=True)
eng = tiled.HipBandEngine(planes, 0.3, [0.001], 100, (0, rows), 0)
drv = tiled.RowTiledSolver(eng)
for rep in range(3):
    eng.reset(); drv.start(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    drv.iterate(100)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"W={W}: enqueue {1e6 * (t1 - t0) / 100:.1f} us/iteration, complete {1e6 * (t2 - t0) / 100:.1f} us/iteration, direct={drv.direct is not None}", flush=True)
dist.destroy_process_group()
