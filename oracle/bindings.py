"""TEST INFRASTRUCTURE — ctypes bindings of the CPU oracle (our C restatement) and, when it
has been built, of the compiled reference (oracle/_ref).  Only tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() import this module."""
import ctypes
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB = os.path.join(_HERE, "_build", "libj2p_oracle.so")
REF_LIB = os.path.join(_HERE, "_ref", "libj2p_ref.so")
REF_OMP_LIB = os.path.join(_HERE, "_ref", "libj2p_ref_omp.so")


def build(ref=True):
    """make the oracle (always) and the reference build (only where /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", _HERE, "oracle"], check=True)
    if ref:
        subprocess.run(["make", "-s", "-C", _HERE, "ref"], check=True)
        # the whole reference program, for the end-to-end CLI test (needs libjpeg/libpng headers)
        subprocess.run(["make", "-s", "-C", _HERE, "ref-cli"], check=False)
        # ... and the same program with compute.o replaced by libjpeg2png_amd.so (the drop-in proof)
        subprocess.run(["make", "-s", "-C", _HERE, "ref-dropin"], check=False)


class _OPlane(ctypes.Structure):
    _fields_ = [("w", ctypes.c_uint), ("h", ctypes.c_uint), ("w_samp", ctypes.c_uint), ("h_samp", ctypes.c_uint),
                ("coef", ctypes.c_void_p), ("quant", ctypes.c_void_p), ("pixels", ctypes.c_void_p)]


_oracle = None
_ref = {}


def oracle_lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_LIB):
            build(ref=False)
        _oracle = ctypes.CDLL(ORACLE_LIB)
        _oracle.oracle_compute.argtypes = [ctypes.c_uint, ctypes.POINTER(_OPlane), ctypes.c_float,
                                           ctypes.POINTER(ctypes.c_float), ctypes.c_uint,
                                           ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]
        _oracle.oracle_decode_plane.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p]
        _oracle.oracle_decode_plane.restype = None
        _oracle.oracle_fdct8x8.argtypes = [ctypes.c_void_p]
        _oracle.oracle_fdct8x8.restype = None
        _oracle.oracle_idct8x8.argtypes = [ctypes.c_void_p]
        _oracle.oracle_idct8x8.restype = None
    return _oracle


def have_ref():
    return os.path.exists(REF_LIB)


def ref_lib(omp=False):
    path = REF_OMP_LIB if omp else REF_LIB
    if path not in _ref:
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not built (needs /root/reference; run make -C oracle ref)")
        lib = ctypes.CDLL(path)
        lib.ref_compute.argtypes = ([ctypes.c_uint] + [ctypes.c_void_p] * 7
                                    + [ctypes.c_float, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p])
        for f in ("dct8x8s", "idct8x8s"):
            getattr(lib, f).argtypes = [ctypes.c_void_p]
            getattr(lib, f).restype = None
        _ref[path] = lib
    return _ref[path]


def canvas_size(planes):
    W = max(p.w * p.w_samp for p in planes)
    H = max(p.h * p.h_samp for p in planes)
    return W, H


def decode_plane(plane):
    """CPU decode_coefficients + unbox (jpeg.c:83-92, box.c:5-19)."""
    lib = oracle_lib()
    d = np.ascontiguousarray(plane.data, dtype=np.int16)
    q = np.ascontiguousarray(plane.quant_table, dtype=np.uint16)
    out = np.empty((plane.h, plane.w), dtype=np.float32)
    lib.oracle_decode_plane(plane.w, plane.h, d.ctypes.data, q.ctypes.data, out.ctypes.data)
    return out


def dct_blocks(blocks, inverse=False, which="oracle"):
    b = np.array(blocks, dtype=np.float32, order="C").reshape(-1, 64)
    if which == "oracle":
        lib = oracle_lib()
        fn = lib.oracle_idct8x8 if inverse else lib.oracle_fdct8x8
    else:
        lib = ref_lib()
        fn = lib.idct8x8s if inverse else lib.dct8x8s
    # the reference assumes 16-byte alignment (ooura/dct.c:35): use an aligned scratch block
    scratch = np.zeros(64 + 8, dtype=np.float32)
    off = (-scratch.ctypes.data % 16) // 4
    blk = scratch[off:off + 64]
    for i in range(b.shape[0]):
        blk[:] = b[i]
        fn(blk.ctypes.data)
        b[i] = blk
    return b


def oracle_compute(planes, weight, pweight, iterations, log=False):
    """Run the C restatement; returns (list of W*H canvas planes, log rows or None)."""
    lib = oracle_lib()
    n = len(planes)
    W, H = canvas_size(planes)
    keep = []
    arr = (_OPlane * n)()
    for i, p in enumerate(planes):
        d = np.ascontiguousarray(p.data, dtype=np.int16)
        q = np.ascontiguousarray(p.quant_table, dtype=np.uint16)
        f = np.ascontiguousarray(p.fdata, dtype=np.float32)
        keep += [d, q, f]
        arr[i] = _OPlane(p.w, p.h, p.w_samp, p.h_samp, d.ctypes.data, q.ctypes.data, f.ctypes.data)
    outs = [np.empty((H, W), dtype=np.float32) for _ in range(n)]
    optr = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
    pw = (ctypes.c_float * n)(*[float(x) for x in pweight])
    rows = np.zeros((max(iterations, 1), 4), dtype=np.float64) if log else None
    rc = lib.oracle_compute(n, arr, float(weight), pw, int(iterations), optr,
                            rows.ctypes.data if log else None)
    if rc != 0:
        raise RuntimeError(f"oracle_compute failed: {rc}")
    return outs, (rows[:iterations] if log else None)


def ref_compute(planes, weight, pweight, iterations, log=False, omp=False):
    """Run the UNMODIFIED reference compute() (compiled from /root/reference).
    Returns (canvas planes, log rows parsed from the reference's CSV or None, seconds in compute())."""
    lib = ref_lib(omp)
    n = len(planes)
    W, H = canvas_size(planes)
    keep = []

    def uarr(vals):
        a = (ctypes.c_uint * n)(*vals)
        keep.append(a)
        return ctypes.cast(a, ctypes.c_void_p)

    def parr(arrays):
        keep.extend(arrays)
        a = (ctypes.c_void_p * n)(*[x.ctypes.data for x in arrays])
        keep.append(a)
        return ctypes.cast(a, ctypes.c_void_p)

    data = [np.ascontiguousarray(p.data, dtype=np.int16) for p in planes]
    fdata = [np.ascontiguousarray(p.fdata, dtype=np.float32) for p in planes]
    quant = [np.ascontiguousarray(p.quant_table, dtype=np.uint16) for p in planes]
    outs = [np.empty((H, W), dtype=np.float32) for _ in range(n)]
    pw = (ctypes.c_float * n)(*[float(x) for x in pweight])
    oW, oH = ctypes.c_uint(), ctypes.c_uint()
    secs = ctypes.c_double()
    csv = None
    if log:
        fd, csv = tempfile.mkstemp(suffix=".csv")
        os.close(fd)
    rc = lib.ref_compute(n, uarr([p.w for p in planes]), uarr([p.h for p in planes]),
                         uarr([p.w_samp for p in planes]), uarr([p.h_samp for p in planes]),
                         parr(data), parr(fdata), parr(quant), float(weight),
                         ctypes.cast(pw, ctypes.c_void_p), int(iterations), parr(outs),
                         ctypes.cast(ctypes.byref(oW), ctypes.c_void_p),
                         ctypes.cast(ctypes.byref(oH), ctypes.c_void_p),
                         csv.encode() if csv else None, ctypes.cast(ctypes.byref(secs), ctypes.c_void_p))
    if rc != 0:
        raise RuntimeError(f"ref_compute failed: {rc}")
    assert (oW.value, oH.value) == (W, H)
    rows = None
    if csv:
        if iterations:
            rows = np.loadtxt(csv, delimiter=",", skiprows=1, usecols=(3, 4, 5, 6), ndmin=2)
        else:
            rows = np.zeros((0, 4))
        os.unlink(csv)
    return outs, rows, secs.value
