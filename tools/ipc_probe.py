"""probe: does hipIpcGetMemHandle / hipIpcOpenMemHandle work between two processes on this box?"""
import ctypes
import multiprocessing as mp
import os
import sys

import numpy as np


class Handle(ctypes.Structure):          # hipIpcMemHandle_t: 64 opaque bytes, passed BY VALUE to Open
    _fields_ = [("reserved", ctypes.c_char * 64)]


def hip():
    h = ctypes.CDLL("libamdhip64.so")
    h.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    h.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    h.hipIpcGetMemHandle.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    h.hipIpcOpenMemHandle.argtypes = [ctypes.POINTER(ctypes.c_void_p), Handle, ctypes.c_uint]
    h.hipIpcCloseMemHandle.argtypes = [ctypes.c_void_p]
    h.hipGetErrorString.restype = ctypes.c_char_p
    return h


def child(handle_bytes, q):
    h = hip()
    assert h.hipSetDevice(0) == 0
    hb = Handle.from_buffer_copy(handle_bytes)
    p = ctypes.c_void_p()
    rc = h.hipIpcOpenMemHandle(ctypes.byref(p), hb, 1)
    if rc != 0:
        q.put("open failed: %d %s" % (rc, h.hipGetErrorString(rc)))
        return
    a = np.arange(1024, dtype=np.float32) * 3
    rc = h.hipMemcpy(p, a.ctypes.data, a.nbytes, 1)
    h.hipDeviceSynchronize()
    h.hipIpcCloseMemHandle(p)
    q.put("wrote rc=%d" % rc)


if __name__ == "__main__":
    mp.set_start_method("spawn")
    h = hip()
    assert h.hipSetDevice(0) == 0
    p = ctypes.c_void_p()
    assert h.hipMalloc(ctypes.byref(p), 4096) == 0
    hb = Handle()
    rc = h.hipIpcGetMemHandle(ctypes.byref(hb), p)
    print("get handle rc", rc, h.hipGetErrorString(rc), "HSA_ENABLE_IPC_MODE_LEGACY=", os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
    if rc != 0:
        sys.exit(1)
    q = mp.Queue()
    pr = mp.Process(target=child, args=(bytes(hb), q))
    pr.start()
    print(q.get(timeout=120))
    pr.join()
    out = np.zeros(1024, dtype=np.float32)
    h.hipMemcpy(out.ctypes.data, p, out.nbytes, 2)
    print("parent sees", out[:4], "ok" if out[5] == 15 else "MISMATCH")
