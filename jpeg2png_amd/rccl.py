"""RCCL straight on the solver's own HIP streams (no torch.distributed in the iteration loop).

The row-tiled driver needs two tiny exchanges per iteration — an all-gather of a few hundred doubles that sits
on the critical path and a 2-row halo send/recv that does not.  Through torch.distributed each of them runs on
ProcessGroupNCCL's internal stream, i.e. between two cross-stream dependencies (~25 us each on this pool) plus
the dispatcher's host cost.  Here the same librccl the process already has is called through ctypes with the
stream the neighbouring kernels run on, so the all-gather is just one more kernel in the solver's stream.

torch.distributed is still used once, to hand rank 0's ncclUniqueId to the other ranks.
"""
import ctypes
import os

NCCL_UNIQUE_ID_BYTES = 128           # rccl.h:40
ncclFloat32, ncclFloat64 = 7, 8      # rccl.h ncclDataType_t


class _UniqueId(ctypes.Structure):   # passed BY VALUE to ncclCommInitRank (rccl.h:43,220)
    _fields_ = [("internal", ctypes.c_char * NCCL_UNIQUE_ID_BYTES)]


class RcclError(RuntimeError):
    pass


_lib = None


def _load():
    """the librccl this process is already bound to (torch's bundled copy when torch is loaded), else ROCm's"""
    global _lib
    if _lib is not None:
        return _lib
    path = None
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "librccl" in line:
                    path = line.split()[-1]
                    break
    except OSError:
        pass
    if path is None:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so") if spec and spec.origin else None
        path = cand if cand and os.path.exists(cand) else "librccl.so"
    lib = ctypes.CDLL(path)
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    lib.ncclGetErrorString.argtypes = [ctypes.c_int]
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
    lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    lib.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                  ctypes.c_void_p]
    lib.ncclSend.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.ncclRecv.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        raise RcclError(f"{what}: {_load().ncclGetErrorString(rc).decode()} ({rc})")


class Communicator:
    """one RCCL communicator over the ranks of a torch.distributed group (one process per GPU)"""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        lib = _load()
        self._lib = lib
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        box = [None]
        if self.rank == 0:
            uid = _UniqueId()
            _check(lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
            box = [ctypes.string_at(ctypes.addressof(uid), NCCL_UNIQUE_ID_BYTES)]   # all 128 raw bytes
        if self.world > 1:
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group, device=torch.device("cuda", torch.cuda.current_device()))
        uid = _UniqueId.from_buffer_copy(box[0])
        self._comm = ctypes.c_void_p()
        _check(lib.ncclCommInitRank(ctypes.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")

    def all_gather(self, send_ptr, recv_ptr, count, stream, dtype=ncclFloat64):
        """recv[r*count:(r+1)*count] = rank r's send[0:count]; enqueued on hipStream_t `stream`"""
        _check(self._lib.ncclAllGather(send_ptr, recv_ptr, count, dtype, self._comm, stream), "ncclAllGather")

    def exchange(self, sends, recvs, stream, dtype=ncclFloat32):
        """one grouped launch of ncclSend (ptr, count, peer) and ncclRecv (ptr, count, peer) operations"""
        lib = self._lib
        _check(lib.ncclGroupStart(), "ncclGroupStart")
        try:
            for ptr, count, peer in sends:
                _check(lib.ncclSend(ptr, count, dtype, peer, self._comm, stream), "ncclSend")
            for ptr, count, peer in recvs:
                _check(lib.ncclRecv(ptr, count, dtype, peer, self._comm, stream), "ncclRecv")
        finally:
            _check(lib.ncclGroupEnd(), "ncclGroupEnd")

    def close(self):
        if self._comm:
            self._lib.ncclCommDestroy(self._comm)
            self._comm = ctypes.c_void_p()
