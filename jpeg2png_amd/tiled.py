"""Row tiling of ONE plane set over several GPUs (BASELINE.json config 4).

One process per GPU.  Each rank owns a contiguous band of canvas rows (a multiple
of 8*h_samp and of the 16-row gradient tile, so no DCT block and no gradient tile
straddles two GPUs) and runs the same two phase kernels as the single-GPU solver.
Per iteration there are exactly two exchanges:

  1. after phase_gradient: all-gather of one double per channel per row-of-tiles
     (sum g*g).  Every rank then reduces the SAME global array in the SAME fixed
     order inside phase_project, so ||g|| — and therefore the result — does not
     depend on the number of GPUs;
  2. after phase_project: the first/last 2 rows of the new iterate go to the
     neighbouring bands' halo rows (TGV2 reaches 2 rows, compute.c:137-143,165-183;
     x_{k-1}'s halo is already there from the previous iteration, so each rank
     forms the FISTA point of its halo locally).  Grouped isend/irecv, no wrap-around.

torch.distributed is plumbing only (backend "nccl" = RCCL over xGMI on the GPUs,
"gloo" in the CPU tests); the arithmetic lives behind the `engine` object.  The
production engine is HipBandEngine (the C-ABI solver, device memory aliased as
torch tensors); tests drive the same exchange code with a CPU engine over gloo.
On GPUs the two exchanges go to librccl directly (jpeg2png_amd/rccl.py) on the
solver's own streams; torch.distributed then only bootstraps the communicators.
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import J2P_HALO_ROWS, J2P_MAX_CHANNELS, J2P_TILE_ROWS, Solver, log_rows_from_sums


def band_alignment(planes):
    """rows a band boundary must be a multiple of: lcm(16, 8*h_samp of every channel)."""
    import math
    a = J2P_TILE_ROWS
    for p in planes:
        a = math.lcm(a, 8 * p.h_samp)
    return a


def split_rows(H, world, align):
    """contiguous, aligned, near-equal bands [(row_begin, row_end)] covering [0, H)."""
    units = (H + align - 1) // align
    if units < world:
        raise ValueError(f"canvas of {H} rows has only {units} bands of {align} rows for {world} ranks")
    out, start = [], 0
    for r in range(world):
        n = units // world + (1 if r < units % world else 0)
        end = min(H, (start + n) * align) if r < world - 1 else H
        out.append((start * align, end))
        start += n
    return out


class _DeviceArray:
    """__cuda_array_interface__ view of raw device memory (no copy, no ownership)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def alias_tensor(ptr, n, dtype, device):
    typestr = {torch.float32: "<f4", torch.float64: "<f8"}[dtype]
    t = torch.as_tensor(_DeviceArray(ptr, n, typestr), device=device)
    assert t.data_ptr() == int(ptr) and t.dtype == dtype
    return t


class HipBandEngine:
    """The C-ABI band solver with its exchange buffers exposed as torch tensors."""

    def __init__(self, planes, weight, pweight, iterations, band, device, band_local_arrays=True):
        self.device = torch.device("cuda", device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.solver = Solver(planes, weight, pweight, iterations, device=device,
                             stream=ctypes.c_void_p(self.stream.cuda_stream), band=band,
                             band_local_arrays=band_local_arrays)
        self.nch = self.solver.nch
        self.weight, self.pweight = float(weight), [float(x) for x in pweight]
        self.log_local = None
        e = self.solver.exchange_info()
        self.local_tile_rows, self.global_tile_rows = e.local_tile_rows, e.global_tile_rows
        self.first_tile_row = e.first_tile_row
        self.partials_local = alias_tensor(e.partials_local, e.local_tile_rows * self.nch, torch.float64, self.device)
        self.partials_all = alias_tensor(e.partials_all, e.global_tile_rows * self.nch, torch.float64, self.device)
        self._halo = {}
        self._parity = 0
        self.side = torch.cuda.Stream(device=self.device)     # edge segments of the gradient phase
        self.comm = torch.cuda.Stream(device=self.device)     # halo exchange
        self._edges_done = None

    def enable_logging(self):
        """the phase calls also produce the band's tv / tv2 / prob sums (5 doubles, aliased as a tensor)"""
        self.solver.set_logging(True)
        e = self.solver.exchange_info()
        self.log_local = alias_tensor(e.log_local, 2 + J2P_MAX_CHANNELS, torch.float64, self.device)

    def _views(self):
        if self._parity not in self._halo:
            e = self.solver.exchange_info()
            n = e.halo_floats
            mk = lambda p: alias_tensor(p, n, torch.float32, self.device)  # noqa: E731
            self._halo[self._parity] = {k: [mk(getattr(e, k)[c]) for c in range(self.nch)]
                                        for k in ("send_top", "recv_top", "send_bottom", "recv_bottom")}
        return self._halo[self._parity]

    def halo(self):
        return self._views()

    def stream_context(self):
        """collectives order themselves against torch's CURRENT stream: make it the solver's"""
        return torch.cuda.stream(self.stream)

    def phase_gradient(self):
        self.solver.phase_gradient()

    def phase_project(self):
        self.solver.phase_project()
        self._parity ^= 1           # the iterate now lives in the other buffer

    # -- gradient phase split in two, to overlap the halo exchange with compute ----------------
    @property
    def can_split(self):
        """the split phases answer only in the experiments build of the library (the release library returns
        J2P_ESTATE from the *_part calls: measured slower than whole phases)"""
        import jpeg2png_amd as j
        return j.experiments_build() and (self.solver.row_end - self.solver.row_begin) >= 3 * J2P_TILE_ROWS

    def gradient_interior(self):
        """all segments but the band's first and last: no halo row is read (solver stream)"""
        self.solver.phase_gradient_part(1)

    def gradient_edges(self, halo_ready=None):
        """first/last segment on a side stream, after `halo_ready` (event recorded behind the
        arrival of the neighbours' rows); returns nothing, finish_gradient() joins"""
        if halo_ready is not None:
            self.side.wait_event(halo_ready)
        else:
            self.side.wait_stream(self.stream)
        self.solver.phase_gradient_part(2, ctypes.c_void_p(self.side.cuda_stream))
        self._edges_done = self.side.record_event()

    def finish_gradient(self):
        if self._edges_done is not None:
            self.stream.wait_event(self._edges_done)
            self._edges_done = None
        self.solver.phase_rowsums()

    def gradient_edges_inline(self, halo_ready=None):
        """first/last segment on the solver's own stream, behind `halo_ready`"""
        if halo_ready is not None:
            self.stream.wait_event(halo_ready)
        self.solver.phase_gradient_part(2)

    # -- projection phase split in two, so that the halo rows leave as early as possible ---------
    def project_boundary(self):
        """norm + the band's first and last block rows; halo() then refers to the NEW iterate"""
        self.solver.phase_project_part(1)
        self._parity ^= 1

    def project_interior(self):
        self.solver.phase_project_part(2)

    def project_done_event(self):
        return self.stream.record_event()

    def comm_context(self, after):
        """context in which the halo exchange is issued: the comm stream, behind the event `after`"""
        self.comm.wait_event(after)
        return torch.cuda.stream(self.comm)

    def comm_done_event(self):
        return self.comm.record_event()

    def wait_halo(self, event):
        self.stream.wait_event(event)

    def commit_initial_halo(self):
        self.solver.commit_initial_halo()

    def reset(self):
        self.solver.reset()
        self._parity = 0

    def download(self, c):
        return self.solver.download(c)

    def close(self):
        self.solver.close()


class RowTiledSolver:
    """Drives one band engine per rank through the iteration loop (compute.c:427-453).

    overlap=True (default where the engine supports it): the projection does the band's first and last
    block rows first, the halo exchange then runs on its own stream while the rest of the projection and
    the interior of the next gradient phase compute; only the band's first and last 16-row gradient
    segments wait for the neighbours' rows.  The all-gather of the norm partials stays on the critical
    path (it is the global dependency of the algorithm).  log=True also gathers the bands' tv / tv2 /
    prob sums every iteration (log_rows() returns the reference's CSV values)."""

    def __init__(self, engine, group=None, overlap=True, log=False):
        self.e = engine
        self.group = group
        self.log = bool(log)
        self._logbuf = None
        self._logged = 0
        if self.log:
            engine.enable_logging()
        self.overlap = bool(overlap) and getattr(engine, "can_split", False)
        self._halo_ready = None
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._ops = {}
        self.up = self.rank - 1 if self.rank > 0 else None
        self.down = self.rank + 1 if self.rank < self.world - 1 else None
        # debugging / timing aid for single-GPU boxes: a lone rank exchanges its halo rows with ITSELF and
        # runs the real all-gather, so the whole communication path (grouped send/recv on the aliased
        # tensors, stream choreography, per-iteration host cost) is exercised.  Harmless for the result:
        # the rows received land above the first / below the last image row, which the kernels mask.
        self.self_neighbours = self.world == 1 and os.environ.get("J2P_TILED_SELF_NEIGHBOURS", "0") == "1"
        if self.self_neighbours:
            self.up = self.down = self.rank
        # RCCL called directly on the solver's streams where possible (GPU engine, nccl backend); the
        # torch.distributed calls remain for CPU engines (gloo tests), unequal bands and as the fallback
        self.direct = None
        want = os.environ.get("J2P_TILED_TRANSPORT", "rccl")
        if want == "rccl" and getattr(engine, "can_split", None) is not None and dist.get_backend(group) == "nccl":
            try:
                from . import rccl
                with torch.cuda.device(engine.device):
                    # two communicators: RCCL orders the operations of ONE communicator among themselves even
                    # across streams, and the halo exchange must not get in the all-gather's way
                    self.direct = rccl.Communicator(group)
                    self.direct_p2p = rccl.Communicator(group)
            except Exception as ex:                       # noqa: BLE001 — any failure means "use torch's path"
                import sys
                print(f"jpeg2png_amd.tiled: direct RCCL unavailable ({ex}); using torch.distributed", file=sys.stderr)
        # a CPU-only process group (gloo) under a GPU engine: both exchanges are staged through host memory — several
        # processes on real kernels without RCCL (two ranks sharing one GPU in tests/test_tiled_multiprocess_gpu.py)
        self.host_staged = dist.get_backend(group) == "gloo" and engine.partials_local.device.type != "cpu"
        counts = torch.zeros(self.world, dtype=torch.int64)
        counts[self.rank] = engine.local_tile_rows
        if not self.host_staged:
            counts = counts.to(engine.partials_local.device)
        dist.all_reduce(counts, group=group)
        self.counts = [int(x) for x in counts.cpu()]
        self.equal = len(set(self.counts)) == 1
        if not self.equal and not self.host_staged:
            self._stage = torch.zeros(self.world * max(self.counts) * engine.nch, dtype=torch.float64,
                                      device=engine.partials_local.device)
            self._pad = torch.zeros(max(self.counts) * engine.nch, dtype=torch.float64,
                                    device=engine.partials_local.device)

    # -- exchanges ---------------------------------------------------------
    def gather_partials(self):
        e = self.e
        if self.world == 1 and not self.self_neighbours:
            if e.partials_all.data_ptr() != e.partials_local.data_ptr():
                e.partials_all.copy_(e.partials_local)
            return
        if self.host_staged:
            # device -> host (synchronises the solver's stream), gloo all-gather of equal-size padded pieces, host -> device
            m = max(self.counts) * e.nch
            mine = torch.zeros(m, dtype=torch.float64)
            mine[: e.partials_local.numel()] = e.partials_local.cpu()
            pieces = [torch.zeros(m, dtype=torch.float64) for _ in range(self.world)]
            dist.all_gather(pieces, mine, group=self.group)
            e.partials_all.copy_(torch.cat([pieces[r][: self.counts[r] * e.nch] for r in range(self.world)]))
            return
        if self.equal and self.direct is not None:
            self.direct.all_gather(e.partials_local.data_ptr(), e.partials_all.data_ptr(), e.partials_local.numel(),
                                   ctypes.c_void_p(torch.cuda.current_stream(e.device).cuda_stream))
            return
        if self.equal:
            dist.all_gather_into_tensor(e.partials_all, e.partials_local, group=self.group)
            return
        m = max(self.counts) * e.nch
        self._pad.zero_()
        self._pad[: e.partials_local.numel()] = e.partials_local
        dist.all_gather_into_tensor(self._stage, self._pad, group=self.group)
        off = 0
        for r, n in enumerate(self.counts):
            e.partials_all[off: off + n * e.nch] = self._stage[r * m: r * m + n * e.nch]
            off += n * e.nch

    # -- CSV log sums (the "+3 doubles" of the norm exchange; only when logging) ------------------
    def _gather_log(self):
        """all-gather the band's {tv, tv2, prob per channel} of the iteration just finished"""
        e = self.e
        k = 2 + J2P_MAX_CHANNELS
        if self._logbuf is None or self._logged >= self._logbuf.shape[0]:
            grown = torch.zeros((max(64, 2 * self._logged), self.world, k), dtype=torch.float64, device=e.log_local.device)
            if self._logbuf is not None:
                grown[: self._logged] = self._logbuf[: self._logged]
            self._logbuf = grown
        dst = self._logbuf[self._logged]
        if self.host_staged:
            pieces = [torch.zeros(k, dtype=torch.float64) for _ in range(self.world)]
            dist.all_gather(pieces, e.log_local.cpu(), group=self.group)
            dst.copy_(torch.stack(pieces))
        elif self.world == 1 and not self.self_neighbours:
            dst[0].copy_(e.log_local)
        elif self.direct is not None:
            self.direct.all_gather(e.log_local.data_ptr(), dst.data_ptr(), k,
                                   ctypes.c_void_p(torch.cuda.current_stream(e.device).cuda_stream))
        else:
            dist.all_gather_into_tensor(dst.view(-1), e.log_local, group=self.group)
        self._logged += 1

    def log_rows(self):
        """the reference's CSV values [objective, prob_dist, tv, tv2] of every iteration run so far
        (bands added up in rank order; synchronises)"""
        if not self.log:
            raise RuntimeError("RowTiledSolver was created without log=True")
        sums = self._logbuf[: self._logged].cpu().numpy() if self._logged else torch.zeros((0, self.world, 5)).numpy()
        total = sums[:, 0, :].copy()
        for r in range(1, self.world):
            total += sums[:, r, :]
        return log_rows_from_sums(self.e.nch, self.e.weight, self.e.pweight, total)

    def exchange_halo(self):
        if self.up is None and self.down is None:
            return
        h = self.e.halo()
        key = tuple(t.data_ptr() for k in ("send_top", "recv_top", "send_bottom", "recv_bottom") for t in h[k])
        ops = self._ops.get(key)
        if ops is None:
            # one grouped send/recv per neighbour and channel; built once per iterate buffer (the
            # iterate ping-pongs between two buffers, so there are two op lists)
            ops = []
            for c in range(self.e.nch):
                if self.up is not None:
                    ops.append(dist.P2POp(dist.isend, h["send_top"][c], self.up, self.group))
                    ops.append(dist.P2POp(dist.irecv, h["recv_top"][c], self.up, self.group))
                if self.down is not None:
                    ops.append(dist.P2POp(dist.isend, h["send_bottom"][c], self.down, self.group))
                    ops.append(dist.P2POp(dist.irecv, h["recv_bottom"][c], self.down, self.group))
            self._ops[key] = (ops, h)     # keep the views alive
        else:
            ops = ops[0]
        if self.host_staged:
            host_ops, landing = [], []
            for op in ops:
                if op.op is dist.isend:
                    host_ops.append(dist.P2POp(dist.isend, op.tensor.cpu(), op.peer, self.group))
                else:
                    buf = torch.empty(op.tensor.numel(), dtype=op.tensor.dtype)
                    landing.append((op.tensor, buf))
                    host_ops.append(dist.P2POp(dist.irecv, buf, op.peer, self.group))
            for w in dist.batch_isend_irecv(host_ops):
                w.wait()
            for dev, buf in landing:
                dev.copy_(buf)
            return
        if self.direct is not None:
            sends = [(op.tensor.data_ptr(), op.tensor.numel(), op.peer) for op in ops if op.op is dist.isend]
            recvs = [(op.tensor.data_ptr(), op.tensor.numel(), op.peer) for op in ops if op.op is dist.irecv]
            self.direct_p2p.exchange(sends, recvs, ctypes.c_void_p(torch.cuda.current_stream(self.e.device).cuda_stream))
            return
        for w in dist.batch_isend_irecv(ops):
            w.wait()

    def close(self):
        """release the direct RCCL communicators (the engine is closed by its owner)"""
        if self.direct is not None:
            self.direct.close()
            self.direct_p2p.close()
            self.direct = None

    # -- loop --------------------------------------------------------------
    def start(self):
        """initial halo rows of x_0 (needed when every rank only uploaded its own band)."""
        self._logged = 0
        with self.e.stream_context():
            self.exchange_halo()
            self.e.commit_initial_halo()
        self._halo_ready = None            # the solver's stream itself is behind the exchange

    def _exchange_halo_async(self):
        """halo exchange on the engine's comm stream, behind everything issued so far on the
        solver's stream; returns the event that marks the arrival of the neighbours' rows"""
        e = self.e
        done = e.project_done_event()
        with e.comm_context(done):
            self.exchange_halo()
            return e.comm_done_event()

    def iterate(self, n):
        e = self.e
        if not self.overlap:
            with e.stream_context():
                for _ in range(n):
                    e.phase_gradient()
                    self.gather_partials()
                    e.phase_project()
                    if self.log:
                        self._gather_log()
                    self.exchange_halo()
            return
        # One stream carries the whole iteration; only the halo exchange runs beside it.  The rows the
        # neighbours need are projected first, so the exchange has the interior of the projection AND the
        # interior of the next gradient phase (two kernels, ~2/3 of the iteration) to complete in; the
        # all-gather of the norm partials is the only communication left on the critical path.
        for _ in range(n):
            with e.stream_context():
                e.gradient_interior()
                e.gradient_edges_inline(self._halo_ready)
                e.finish_gradient()
                self.gather_partials()
                e.project_boundary()
            self._halo_ready = self._exchange_halo_async()
            with e.stream_context():
                e.project_interior()
                if self.log:
                    self._gather_log()
        # leave the solver's stream consistent for whoever comes next (download, reset, ...)
        if self._halo_ready is not None:
            e.wait_halo(self._halo_ready)
