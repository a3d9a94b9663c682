"""jpeg2png_amd — MI355X (gfx950) implementation of jpeg2png's TV/TGV deblocking solver.

Python here is only a ctypes binding of the C-ABI in include/jpeg2png_amd.h (used by
tests/ and bench.py, and as the host-side mirror of the reference's compute()
interface).  The product is libjpeg2png_amd.so: hand-written HIP kernels behind a
plain-C shim plus a C `compute()` with the reference's own signature.

There is NO CPU fallback: if the shared library is missing or no GPU is present the
calls raise.
"""
import ctypes
import os

import numpy as np

from .synth import Plane  # noqa: F401  (re-export: the Python twin of struct coef)

_HERE = os.path.dirname(os.path.abspath(__file__))
# J2P_LIBRARY selects another build of the library (A/B timing of kernel variants on one box)
LIB_PATH = os.environ.get("J2P_LIBRARY") or os.path.join(_HERE, "libjpeg2png_amd.so")

J2P_MAX_CHANNELS = 3
J2P_HALO_ROWS = 2
J2P_TILE_ROWS = 16


class J2PError(RuntimeError):
    pass


class _CPlane(ctypes.Structure):
    _fields_ = [("w", ctypes.c_uint), ("h", ctypes.c_uint),
                ("w_samp", ctypes.c_uint), ("h_samp", ctypes.c_uint),
                ("data", ctypes.c_void_p), ("fdata", ctypes.c_void_p),
                ("quant_table", ctypes.c_void_p)]


class _CBand(ctypes.Structure):
    _fields_ = [("row_begin", ctypes.c_uint), ("row_end", ctypes.c_uint)]


class _CLogRow(ctypes.Structure):
    _fields_ = [("objective", ctypes.c_double), ("prob_dist", ctypes.c_double),
                ("tv", ctypes.c_double), ("tv2", ctypes.c_double)]


_ROWS_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.POINTER(_CLogRow))
_PROGRESS_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_uint)


class _CJob(ctypes.Structure):
    _fields_ = [("nchannel", ctypes.c_uint), ("planes", _CPlane * 3), ("separate", ctypes.c_int),
                ("weight", ctypes.c_float * 3), ("pweight", ctypes.c_float * 3), ("iterations", ctypes.c_uint * 3),
                ("out_bits", ctypes.c_uint), ("out_w", ctypes.c_uint), ("out_h", ctypes.c_uint),
                ("out_rgb", ctypes.c_void_p), ("out_planes", ctypes.c_void_p * 3),
                ("on_rows", _ROWS_CB), ("on_progress", _PROGRESS_CB), ("user", ctypes.c_void_p), ("tile", ctypes.c_int),
                ("tile_first", ctypes.c_uint), ("tile_count", ctypes.c_uint), ("tile_min_band_pixels", ctypes.c_size_t)]


class _CExchange(ctypes.Structure):
    _fields_ = [("partials_local", ctypes.c_void_p), ("local_tile_rows", ctypes.c_uint),
                ("partials_all", ctypes.c_void_p), ("global_tile_rows", ctypes.c_uint),
                ("first_tile_row", ctypes.c_uint),
                ("send_top", ctypes.c_void_p * 3), ("recv_top", ctypes.c_void_p * 3),
                ("send_bottom", ctypes.c_void_p * 3), ("recv_bottom", ctypes.c_void_p * 3),
                ("halo_floats", ctypes.c_size_t), ("log_local", ctypes.c_void_p)]


# every symbol include/jpeg2png_amd.h and include/jpeg2png_amd_compute.h declare
C_ABI_SYMBOLS = [
    "j2p_version", "j2p_last_error", "j2p_device_count",
    "j2p_solver_create", "j2p_solver_destroy", "j2p_solver_canvas", "j2p_solver_band",
    "j2p_solver_reset", "j2p_solver_run", "j2p_solver_phase_gradient", "j2p_solver_phase_project",
    "j2p_solver_phase_gradient_part", "j2p_solver_phase_rowsums", "j2p_solver_phase_project_part",
    "j2p_solver_exchange_info", "j2p_solver_commit_initial_halo", "j2p_solver_download",
    "j2p_solver_download_gradient", "j2p_solver_set_logging", "j2p_log_rows_from_sums",
    "j2p_solver_plane_ptr", "j2p_solver_sync", "j2p_solver_kernel_times", "j2p_solver_enable_timing",
    "j2p_decode_plane", "j2p_dct8x8_blocks", "j2p_math_selftest", "j2p_planes_to_rgb", "j2p_planes_rows_to_rgb", "j2p_sqrt_exhaustive",
    "j2p_pool_trim", "j2p_solver_debug_option", "j2p_solver_stream", "j2p_solver_halo_rows",
    "j2p_solver_norm_from_bands", "j2p_solver_copy_rows", "j2p_solver_alternate_rowsums",
    "j2p_tiled_create", "j2p_tiled_destroy", "j2p_tiled_canvas", "j2p_tiled_band", "j2p_tiled_run", "j2p_tiled_reset", "j2p_tiled_sync",
    "j2p_tiled_download", "j2p_tiled_host_cpu_seconds", "j2p_rccl_version", "j2p_solver_norm_ptr", "j2p_solver_norm_external",
    "j2p_solver_global_rowsums", "j2p_solver_link_bands", "j2p_tiled_exchange",
    "j2p_batch_create", "j2p_batch_destroy", "j2p_batch_submit", "j2p_batch_wait",
    "compute", "j2p_compute", "j2p_compute_tiled", "j2p_compute_timing", "j2p_debug_fail_run_after", "j2p_solver_launches_per_iteration", "j2p_solver_timing_overhead",
    "j2p_debug_build", "j2p_debug_grad_items", "j2p_experiments_build", "j2p_solver_debug_violations", "j2p_solver_trace", "j2p_division_exhaustive",
    "j2p_solver_coefficient_bytes",
]
J2P_OPT_NORM_FOLD, J2P_OPT_JOINT_INWAVE, J2P_OPT_NORM_IN_PROJECT, J2P_OPT_NT_GRADIENT, J2P_OPT_MIXED_PROJECT = 1, 2, 4, 5, 6
J2P_OPT_NARROW_COEFFICIENTS = 7

_lib = None


def build(force=False, verbose=False):
    from .buildlib import build as _build
    return _build(force=force, verbose=verbose)


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  A process must not end up with
    two HIP runtimes (ours from /opt/rocm loaded first, torch's afterwards: torch then reports "No HIP GPUs
    are available"), so when torch is installed but not imported yet, load ITS runtime first; our library's
    libamdhip64.so.N dependency then resolves to that copy, exactly as when torch was imported first."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    bundled = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        try:
            ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def hip_runtime():
    """ctypes handle of the HIP runtime this process (and our library) is bound to — for tests and tools that
    want hipMemcpy & co.  Loading "libamdhip64.so" by name could pull a second copy of the runtime in."""
    load_library()
    with open("/proc/self/maps") as f:
        paths = {line.split()[-1] for line in f if "libamdhip64" in line}
    if not paths:
        raise J2PError("no HIP runtime is mapped into this process")
    return ctypes.CDLL(sorted(paths)[0])


def load_library():
    """dlopen libjpeg2png_amd.so; raises J2PError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    _lib = _bind(LIB_PATH)
    return _lib


class library:
    """context manager: another build of the library for the objects created inside the block — the experiments build
    (-DJ2P_EXPERIMENTS: schedules and environment knobs the release build does not carry, buildlib.build_experiments)
    in the schedule-equivalence tests.  Solvers remember the library they were created from; both copies share the
    process's HIP runtime."""

    def __init__(self, path):
        self.path = path

    def __enter__(self):
        global _lib
        load_library()
        self._saved = _lib
        _lib = _bind(self.path)
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved


_bound = {}


def _bind(path):
    if path in _bound:
        return _bound[path]
    if not os.path.exists(path):
        raise J2PError(f"{path} is missing: run `python -m jpeg2png_amd.buildlib` "
                       "(the HIP extension is the only implementation; there is no fallback)")
    _share_torch_hip_runtime()
    lib = ctypes.CDLL(path)   # RTLD_LOCAL: our `compute` must not interpose other libraries' symbols
    lib.j2p_version.restype = ctypes.c_char_p
    lib.j2p_last_error.restype = ctypes.c_char_p
    lib.j2p_solver_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_uint, ctypes.POINTER(_CPlane), ctypes.c_float,
                                      ctypes.POINTER(ctypes.c_float), ctypes.c_uint, _CBand, ctypes.c_int]
    lib.j2p_solver_destroy.argtypes = [ctypes.c_void_p]
    lib.j2p_solver_destroy.restype = None
    lib.j2p_solver_phase_gradient_part.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.j2p_solver_phase_project_part.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.j2p_solver_set_logging.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.j2p_log_rows_from_sums.argtypes = [ctypes.c_uint, ctypes.c_float, ctypes.POINTER(ctypes.c_float), ctypes.c_uint,
                                           ctypes.c_void_p, ctypes.POINTER(_CLogRow)]
    for name in ("j2p_solver_reset", "j2p_solver_phase_gradient", "j2p_solver_phase_project",
                 "j2p_solver_sync", "j2p_solver_commit_initial_halo", "j2p_solver_phase_rowsums"):
        getattr(lib, name).argtypes = [ctypes.c_void_p]
    lib.j2p_solver_canvas.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)]
    lib.j2p_solver_band.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)]
    lib.j2p_solver_run.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.POINTER(_CLogRow)]
    lib.j2p_solver_exchange_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(_CExchange)]
    lib.j2p_solver_download.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p]
    lib.j2p_solver_download_gradient.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p]
    lib.j2p_solver_plane_ptr.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.POINTER(ctypes.c_void_p)]
    lib.j2p_solver_kernel_times.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                                            ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint)]
    lib.j2p_solver_enable_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.j2p_decode_plane.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p]
    lib.j2p_dct8x8_blocks.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    lib.j2p_device_count.argtypes = [ctypes.POINTER(ctypes.c_int)]
    lib.j2p_math_selftest.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_uint,
                                      ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong)]
    lib.j2p_solver_debug_option.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.j2p_solver_debug_violations.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_uint),
                                                ctypes.POINTER(ctypes.c_ulonglong)]
    lib.j2p_pool_trim.restype = None
    lib.j2p_tiled_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint, ctypes.POINTER(ctypes.c_int),
                                     ctypes.POINTER(ctypes.c_uint), ctypes.c_uint, ctypes.POINTER(_CPlane), ctypes.c_float,
                                     ctypes.POINTER(ctypes.c_float), ctypes.c_uint]
    lib.j2p_tiled_destroy.argtypes = [ctypes.c_void_p]
    lib.j2p_tiled_destroy.restype = None
    lib.j2p_tiled_canvas.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint),
                                     ctypes.POINTER(ctypes.c_uint)]
    lib.j2p_tiled_band.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint),
                                   ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_void_p)]
    lib.j2p_tiled_run.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.POINTER(_CLogRow)]
    lib.j2p_tiled_sync.argtypes = [ctypes.c_void_p]
    lib.j2p_tiled_reset.argtypes = [ctypes.c_void_p]
    lib.j2p_tiled_download.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p]
    lib.j2p_tiled_host_cpu_seconds.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
    lib.j2p_batch_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint, ctypes.POINTER(ctypes.c_int), ctypes.c_uint]
    lib.j2p_batch_destroy.argtypes = [ctypes.c_void_p]
    lib.j2p_batch_destroy.restype = None
    lib.j2p_batch_submit.argtypes = [ctypes.c_void_p, ctypes.POINTER(_CJob), ctypes.POINTER(ctypes.c_int)]
    lib.j2p_batch_wait.argtypes = [ctypes.c_void_p, ctypes.c_int]
    _bound[path] = lib
    return lib


def _check(rc):
    if rc != 0:
        raise J2PError(f"jpeg2png_amd error {rc}: {load_library().j2p_last_error().decode()}")


def debug_build():
    """True when the loaded library was compiled with -DJ2P_DEBUG (address checks in the phase kernels)"""
    return bool(load_library().j2p_debug_build())


def rccl_version():
    """ncclGetVersion of the librccl the C row tiling would dlopen for its `rccl` exchange, or None (no usable library)"""
    v = ctypes.c_int(0)
    return v.value if load_library().j2p_rccl_version(ctypes.byref(v)) == 0 else None


def experiments_build():
    """True when the loaded library is the experiments build (-DJ2P_EXPERIMENTS): the schedules that lost their
    measurements — split phases among them — answer only there"""
    return bool(load_library().j2p_experiments_build())


def device_count():
    n = ctypes.c_int(0)
    _check(load_library().j2p_device_count(ctypes.byref(n)))
    return n.value


def decode_plane(plane, device=0):
    """decode_coefficients + unbox on the GPU (jpeg.c:83-92, box.c:5-19) -> float32 [h, w]."""
    lib = load_library()
    data = np.ascontiguousarray(plane.data, dtype=np.int16)
    q = np.ascontiguousarray(plane.quant_table, dtype=np.uint16)
    out = np.empty((plane.h, plane.w), dtype=np.float32)
    _check(lib.j2p_decode_plane(device, plane.w, plane.h, data.ctypes.data, q.ctypes.data, out.ctypes.data))
    return out


def dct8x8_blocks(blocks, inverse=False, device=0):
    """dct8x8s / idct8x8s (ooura/dct.c:98 / :34) of n blocks of 64 floats on the GPU."""
    lib = load_library()
    b = np.array(blocks, dtype=np.float32, order="C").reshape(-1, 64)
    _check(lib.j2p_dct8x8_blocks(device, b.ctypes.data, b.shape[0], 1 if inverse else 0))
    return b


def math_selftest(n, seed=1, device=0):
    """(division mismatches, sqrt mismatches) of the fast paths vs IEEE on n random operand pairs."""
    d, q = ctypes.c_ulonglong(), ctypes.c_ulonglong()
    _check(load_library().j2p_math_selftest(device, n, seed, ctypes.byref(d), ctypes.byref(q)))
    return d.value, q.value


def sqrt_exhaustive(device=0):
    """(rsq-sequence mismatches, sqrt-sequence mismatches) vs sqrtf() over every float in [2^-100, 2^127)."""
    a, b = ctypes.c_ulonglong(), ctypes.c_ulonglong()
    lib = load_library()
    lib.j2p_sqrt_exhaustive.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong)]
    _check(lib.j2p_sqrt_exhaustive(device, ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value


def division_exhaustive(which, first=0, count=0, device=0):
    """exhaustive checks of the short division, pass 1 / 2 / 3 (include/jpeg2png_amd.h): (mismatches, first offenders)"""
    lib = load_library()
    lib.j2p_division_exhaustive.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.POINTER(ctypes.c_ulonglong)]
    rep = (ctypes.c_ulonglong * 9)()
    _check(lib.j2p_division_exhaustive(device, which, first, count, rep))
    return rep[0], [hex(v) for v in rep[1:] if v]


class Solver:
    """Device-resident working set of one compute() call (include/jpeg2png_amd.h)."""

    def __init__(self, planes, weight, pweight, iterations, device=0, stream=None, band=None,
                 band_local_arrays=False):
        lib = load_library()
        self._lib = lib
        self._h = None
        self.nch = len(planes)
        keep = []
        cpl = (_CPlane * self.nch)()
        for i, p in enumerate(planes):
            d = np.ascontiguousarray(p.data, dtype=np.int16)
            q = np.ascontiguousarray(p.quant_table, dtype=np.uint16)
            f = None if p.fdata is None else np.ascontiguousarray(p.fdata, dtype=np.float32)
            keep += [d, q, f]
            cpl[i] = _CPlane(p.w, p.h, p.w_samp, p.h_samp, d.ctypes.data, None if f is None else f.ctypes.data,
                             q.ctypes.data)
        pw = (ctypes.c_float * self.nch)(*[float(x) for x in pweight])
        b = _CBand(0, 0) if band is None else _CBand(int(band[0]), int(band[1]))
        h = ctypes.c_void_p()
        _check(lib.j2p_solver_create(ctypes.byref(h), device, stream, self.nch, cpl, float(weight), pw,
                                     int(iterations), b, 1 if band_local_arrays else 0))
        self._h = h
        del keep
        W, H = ctypes.c_uint(), ctypes.c_uint()
        _check(lib.j2p_solver_canvas(h, ctypes.byref(W), ctypes.byref(H)))
        self.W, self.H = W.value, H.value
        r0, r1 = ctypes.c_uint(), ctypes.c_uint()
        _check(lib.j2p_solver_band(h, ctypes.byref(r0), ctypes.byref(r1)))
        self.row_begin, self.row_end = r0.value, r1.value

    def close(self):
        if getattr(self, "_h", None) and not getattr(self, "_borrowed", False):
            self._lib.j2p_solver_destroy(self._h)
        self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def reset(self):
        _check(self._lib.j2p_solver_reset(self._h))

    def run(self, n, log=False):
        """n iterations of the loop compute.c:427-453; returns [n,4] (objective, prob_dist, tv, tv2) if log."""
        if not log:
            _check(self._lib.j2p_solver_run(self._h, n, None))
            return None
        rows = (_CLogRow * max(n, 1))()
        _check(self._lib.j2p_solver_run(self._h, n, rows))
        return np.array([[r.objective, r.prob_dist, r.tv, r.tv2] for r in rows[:n]], dtype=np.float64).reshape(n, 4)

    def phase_gradient(self):
        _check(self._lib.j2p_solver_phase_gradient(self._h))

    def phase_project(self):
        _check(self._lib.j2p_solver_phase_project(self._h))

    def phase_gradient_part(self, part, stream=None):
        """part 1 = interior segments (no halo needed), 2 = the band's first/last segment (optionally
        on another hipStream_t); follow part 2 with phase_rowsums() once the solver's stream waits for it."""
        _check(self._lib.j2p_solver_phase_gradient_part(self._h, int(part), stream))

    def phase_project_part(self, part):
        """part 1 = norm + the band's first/last block rows (the rows the neighbours need), 2 = the rest"""
        _check(self._lib.j2p_solver_phase_project_part(self._h, int(part)))

    def set_logging(self, on=True):
        """band solvers: the phase calls also leave the band's tv / tv2 / prob sums in exchange_info().log_local"""
        _check(self._lib.j2p_solver_set_logging(self._h, 1 if on else 0))

    def phase_rowsums(self):
        _check(self._lib.j2p_solver_phase_rowsums(self._h))

    def commit_initial_halo(self):
        _check(self._lib.j2p_solver_commit_initial_halo(self._h))

    def exchange_info(self):
        e = _CExchange()
        _check(self._lib.j2p_solver_exchange_info(self._h, ctypes.byref(e)))
        return e

    def sync(self):
        _check(self._lib.j2p_solver_sync(self._h))

    def download(self, c):
        out = np.empty((self.row_end - self.row_begin, self.W), dtype=np.float32)
        _check(self._lib.j2p_solver_download(self._h, c, out.ctypes.data))
        return out

    def download_gradient(self, c):
        """diagnostics: the objective gradient the last gradient phase wrote for channel c"""
        out = np.empty((self.row_end - self.row_begin, self.W), dtype=np.float32)
        _check(self._lib.j2p_solver_download_gradient(self._h, c, out.ctypes.data))
        return out

    def plane_ptr(self, c):
        p = ctypes.c_void_p()
        _check(self._lib.j2p_solver_plane_ptr(self._h, c, ctypes.byref(p)))
        return p.value

    def launches_per_iteration(self):
        """kernel launches per iteration of an unlogged run (2, or 3 with a reduction launch between the phases)"""
        n = ctypes.c_uint()
        _check(self._lib.j2p_solver_launches_per_iteration(self._h, ctypes.byref(n)))
        return n.value

    def coefficient_bytes(self, c=0):
        """bytes per quantised coefficient the projection reads for channel c: 1 (every |d| <= 127) or 2"""
        n = ctypes.c_uint()
        _check(self._lib.j2p_solver_coefficient_bytes(self._h, int(c), ctypes.byref(n)))
        return n.value

    def debug_option(self, option, value):
        """schedule switches (J2P_OPT_*): speed only, never results"""
        _check(self._lib.j2p_solver_debug_option(self._h, int(option), int(value)))

    def debug_violations(self):
        """J2P_DEBUG builds: (count, first site code, first offset) of the phase kernels' address checks"""
        n, site, off = ctypes.c_ulonglong(), ctypes.c_uint(), ctypes.c_ulonglong()
        _check(self._lib.j2p_solver_debug_violations(self._h, ctypes.byref(n), ctypes.byref(site), ctypes.byref(off)))
        return n.value, site.value, off.value

    def trace(self, on=True, fetch=False, max_records=1 << 19):
        """J2P_TRACE builds (tools/wave_trace.py): switch the per-wavefront records on / off; fetch=True returns the
        records collected so far as an [n, 4] uint64 array (start, first data, end in 10 ns ticks, id)"""
        self._lib.j2p_solver_trace.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint,
                                               ctypes.POINTER(ctypes.c_uint)]
        if not fetch:
            _check(self._lib.j2p_solver_trace(self._h, 1 if on else 0, None, 0, None))
            return None
        out = np.zeros((max_records, 4), dtype=np.uint64)
        n = ctypes.c_uint()
        _check(self._lib.j2p_solver_trace(self._h, 1 if on else 0, out.ctypes.data, max_records, ctypes.byref(n)))
        return out[: n.value]

    def enable_timing(self, every=1):
        """record HIP events around the two phase kernels of every `every`-th iteration (0 = off)."""
        _check(self._lib.j2p_solver_enable_timing(self._h, int(every)))

    def timing_overhead_ms(self):
        """what a bracket of two event records measures by itself on this solver's stream (the scale of what the brackets add to kernel_times())"""
        v = ctypes.c_double()
        _check(self._lib.j2p_solver_timing_overhead(self._h, ctypes.byref(v)))
        return v.value

    def kernel_times(self):
        g, p, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_uint()
        _check(self._lib.j2p_solver_kernel_times(self._h, ctypes.byref(g), ctypes.byref(p), ctypes.byref(n)))
        return g.value, p.value, n.value


def _c_planes(planes):
    keep = []
    cpl = (_CPlane * len(planes))()
    for i, p in enumerate(planes):
        d = np.ascontiguousarray(p.data, dtype=np.int16)
        q = np.ascontiguousarray(p.quant_table, dtype=np.uint16)
        f = None if p.fdata is None else np.ascontiguousarray(p.fdata, dtype=np.float32)
        keep += [d, q, f]
        cpl[i] = _CPlane(p.w, p.h, p.w_samp, p.h_samp, d.ctypes.data, None if f is None else f.ctypes.data, q.ctypes.data)
    return cpl, keep


class TiledSolver:
    """One plane set cut into row bands, band i on devices[i] (ids may repeat), driven from C by one host thread
    per band (include/jpeg2png_amd.h: j2p_tiled_*).  cuts = band boundaries or None for near-equal bands."""

    def __init__(self, planes, weight, pweight, iterations, devices, cuts=None):
        lib = load_library()
        self._lib = lib
        self._h = None
        self.nch = len(planes)
        cpl, keep = _c_planes(planes)
        n = len(devices)
        devs = (ctypes.c_int * n)(*[int(d) for d in devices])
        ccuts = None if cuts is None else (ctypes.c_uint * (n + 1))(*[int(c) for c in cuts])
        pw = (ctypes.c_float * self.nch)(*[float(x) for x in pweight])
        h = ctypes.c_void_p()
        _check(lib.j2p_tiled_create(ctypes.byref(h), n, devs, ccuts, self.nch, cpl, float(weight), pw, int(iterations)))
        self._h = h
        del keep
        W, H, nb = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
        _check(lib.j2p_tiled_canvas(h, ctypes.byref(W), ctypes.byref(H), ctypes.byref(nb)))
        self.W, self.H, self.nband = W.value, H.value, nb.value

    def bands(self):
        out = []
        for b in range(self.nband):
            d, r0, r1 = ctypes.c_int(), ctypes.c_uint(), ctypes.c_uint()
            _check(self._lib.j2p_tiled_band(self._h, b, ctypes.byref(d), ctypes.byref(r0), ctypes.byref(r1), None))
            out.append((d.value, r0.value, r1.value))
        return out

    def run(self, n, log=False):
        if not log:
            _check(self._lib.j2p_tiled_run(self._h, n, None))
            return None
        rows = (_CLogRow * max(n, 1))()
        _check(self._lib.j2p_tiled_run(self._h, n, rows))
        return np.array([[r.objective, r.prob_dist, r.tv, r.tv2] for r in rows[:n]], dtype=np.float64).reshape(n, 4)

    def sync(self):
        _check(self._lib.j2p_tiled_sync(self._h))

    def reset(self):
        _check(self._lib.j2p_tiled_reset(self._h))

    def exchange(self):
        """how the bands exchange row sums and edge rows: "direct", "copy", "rccl" ("none": one plain band)"""
        name = ctypes.c_char_p()
        _check(self._lib.j2p_tiled_exchange(self._h, ctypes.byref(name)))
        return name.value.decode()

    def host_cpu_seconds(self):
        """user + system time the band threads have spent issuing work (they sleep while waiting for each other)"""
        v = ctypes.c_double()
        _check(self._lib.j2p_tiled_host_cpu_seconds(self._h, ctypes.byref(v)))
        return v.value

    def band_solver(self, b):
        """borrowed handle of band b's j2p_solver (kernel timing in bench.py); owned by the TiledSolver"""
        h = ctypes.c_void_p()
        _check(self._lib.j2p_tiled_band(self._h, b, None, None, None, ctypes.byref(h)))
        s = Solver.__new__(Solver)
        s._lib, s._h, s._borrowed = self._lib, h, True
        return s

    def download(self, c):
        out = np.empty((self.H, self.W), dtype=np.float32)
        _check(self._lib.j2p_tiled_download(self._h, c, out.ctypes.data))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.j2p_tiled_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class Batch:
    """Images in flight over slots_per_device worker threads per GPU (include/jpeg2png_amd.h: j2p_batch_*).
    submit() returns a ticket; wait(ticket) returns the job's output: RGB samples [h, w, 3] (bits 8 / 16) or the
    list of float canvas planes (bits 0)."""

    def __init__(self, devices=(0,), slots_per_device=3):
        lib = load_library()
        self._lib = lib
        self._h = None
        devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        h = ctypes.c_void_p()
        _check(lib.j2p_batch_create(ctypes.byref(h), len(devices), devs, int(slots_per_device)))
        self._h = h
        self._pending = {}

    def submit(self, planes, weight, pweight, iterations, separate=False, width=None, height=None, bits=0, tile=False,
               tile_devices=None, tile_min_band_pixels=None, out=None, on_progress=None):
        """tile=True: the image is row-tiled over the batch's devices instead of solved on one of them;
        tile_devices=(first, count): over that slice of the batch's device list only; on_progress(n): called from the worker
        thread whenever n more iterations of one of the job's solves have finished (the CLI's progress bar, jpeg2png.c:449-452)"""
        n = len(planes)
        job = _CJob()
        job.nchannel = n
        job.tile = 1 if tile else 0
        if tile_devices:
            job.tile_first, job.tile_count = int(tile_devices[0]), int(tile_devices[1])
        if tile_min_band_pixels is not None:           # 0 = no gate at all (tests with small images)
            job.tile_min_band_pixels = int(tile_min_band_pixels) if tile_min_band_pixels else ctypes.c_size_t(-1).value
        cpl, keep = _c_planes(planes)
        for c in range(n):
            job.planes[c] = cpl[c]
        job.separate = 1 if separate else 0
        ws = list(weight) if isinstance(weight, (list, tuple)) else [weight] * n
        its = list(iterations) if isinstance(iterations, (list, tuple)) else [iterations] * n
        for c in range(n):
            job.weight[c], job.pweight[c], job.iterations[c] = float(ws[c]), float(pweight[c]), int(its[c])
        # canvas of a compute() call (compute.c:410-416): all components' for a joint solve, its own for each of the
        # separate calls of `-s` (jpeg2png.c:147-152)
        W = max(p.w * p.w_samp for p in planes)
        H = max(p.h * p.h_samp for p in planes)
        shapes = [(p.h * p.h_samp, p.w * p.w_samp) if separate else (H, W) for p in planes]
        if bits:
            # (out: a caller's own RGB array, reused between jobs — nothing is then mapped or faulted in while other jobs'
            # kernels run, which costs those a stalled launch each time)
            # (the C side writes height * width * 3 samples of bits / 8 bytes: anything else is a heap overflow or a
            # half-written array, so it is an error here, not an assert that -O strips)
            if bits not in (8, 16):
                raise J2PError("job: out_bits must be 0, 8 or 16")
            if out is None:
                out = np.empty((height, width, 3), dtype=np.uint8 if bits == 8 else ">u2")
            if not (isinstance(out, np.ndarray) and out.shape == (height, width, 3) and out.flags["C_CONTIGUOUS"]
                    and out.flags["WRITEABLE"] and out.dtype.itemsize == bits // 8 and out.dtype.kind in "ui"):
                raise J2PError(f"out must be a writeable C-contiguous ({height}, {width}, 3) array of "
                                           f"{bits // 8}-byte integers for bits = {bits}")
            job.out_bits, job.out_w, job.out_h = bits, width, height
            job.out_rgb = out.ctypes.data
        else:
            out = [np.empty(shapes[c], dtype=np.float32) for c in range(n)]
            for c in range(n):
                job.out_planes[c] = out[c].ctypes.data
        if on_progress is not None:
            cb = _PROGRESS_CB(lambda _user, n: on_progress(int(n)))
            job.on_progress = cb
            keep = (keep, cb)                       # the trampoline lives as long as the job
        t = ctypes.c_int()
        _check(self._lib.j2p_batch_submit(self._h, ctypes.byref(job), ctypes.byref(t)))
        self._pending[t.value] = (out, keep)
        return t.value

    def wait(self, ticket):
        out, _keep = self._pending.pop(ticket)
        _check(self._lib.j2p_batch_wait(self._h, ticket))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.j2p_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def log_rows_from_sums(nch, weight, pweight, sums):
    """log rows [objective, prob_dist, tv, tv2] from per-iteration global sums (n x 5: tv, tv2, prob per channel)"""
    lib = load_library()
    sums = np.ascontiguousarray(sums, dtype=np.float64).reshape(-1, 2 + J2P_MAX_CHANNELS)
    n = sums.shape[0]
    rows = (_CLogRow * max(n, 1))()
    pw = (ctypes.c_float * nch)(*[float(x) for x in pweight])
    _check(lib.j2p_log_rows_from_sums(nch, float(weight), pw, n, sums.ctypes.data, rows))
    return np.array([[r.objective, r.prob_dist, r.tv, r.tv2] for r in rows[:n]], dtype=np.float64).reshape(n, 4)


class _CCoef(ctypes.Structure):
    """struct coef (jpeg2png.h:7-20, restated in include/jpeg2png_amd_compute.h)"""
    _fields_ = [("h", ctypes.c_uint), ("w", ctypes.c_uint), ("h_samp", ctypes.c_uint), ("w_samp", ctypes.c_uint),
                ("data", ctypes.c_void_p), ("fdata", ctypes.c_void_p), ("quant_table", ctypes.c_uint16 * 64)]


class _CComputeTimes(ctypes.Structure):
    _fields_ = [(k, ctypes.c_double) for k in ("create_ms", "issue_ms", "housekeeping_ms", "wait_ms", "download_ms", "destroy_ms", "total_ms")]


def compute_c(planes, weight, pweight, iterations, device=0, repeat=1, splits=None):
    """The C drop-in itself — j2p_compute(), what compute() (compute.h:8) is behind its die() wrapper — called the way
    the reference's decode_file() calls it (jpeg2png.c:141-152): planes that libc allocated (alloc_simd, utils.h:89-98),
    the float plane freed and a new one handed back (compute.c:304-305, 455-461), no logger, no progress bar.
    Returns (canvas planes of the last call, [seconds inside j2p_compute per call]): the host-to-host cost of the
    boundary, pageable memory on both sides.  `splits` (a list) receives one dict per call from j2p_compute_timing():
    where that call's wall time went."""
    import time
    lib = load_library()
    libc = ctypes.CDLL(None)
    libc.aligned_alloc.restype = ctypes.c_void_p
    libc.aligned_alloc.argtypes = [ctypes.c_size_t, ctypes.c_size_t]
    libc.malloc.restype = ctypes.c_void_p
    libc.malloc.argtypes = [ctypes.c_size_t]
    libc.free.argtypes = [ctypes.c_void_p]
    lib.j2p_compute.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float,
                                ctypes.c_void_p, ctypes.c_uint]
    n = len(planes)
    pw = (ctypes.c_float * n)(*[float(x) for x in pweight])
    seconds, outs = [], None
    for _ in range(repeat):
        coefs = (_CCoef * n)()
        for c, p in enumerate(planes):
            if p.fdata is None:
                raise J2PError("compute_c() expects decoded planes in fdata (jpeg.c:83-92); use decode_plane()")
            d = np.ascontiguousarray(p.data, dtype=np.int16)
            f = np.ascontiguousarray(p.fdata, dtype=np.float32)
            coefs[c].w, coefs[c].h, coefs[c].w_samp, coefs[c].h_samp = p.w, p.h, p.w_samp, p.h_samp
            coefs[c].data = libc.malloc(d.nbytes)
            coefs[c].fdata = libc.aligned_alloc(16, (f.nbytes + 15) & ~15)
            ctypes.memmove(coefs[c].data, d.ctypes.data, d.nbytes)
            ctypes.memmove(coefs[c].fdata, f.ctypes.data, f.nbytes)
            for k, q in enumerate(np.asarray(p.quant_table, dtype=np.uint16).reshape(64)):
                coefs[c].quant_table[k] = int(q)
        t0 = time.perf_counter()
        rc = lib.j2p_compute(int(device), n, coefs, None, None, float(weight), pw, int(iterations))
        seconds.append(time.perf_counter() - t0)
        if splits is not None and rc == 0:
            ct = _CComputeTimes()
            if lib.j2p_compute_timing(ctypes.byref(ct)) == 0:
                splits.append({k: getattr(ct, k) for k, _ in _CComputeTimes._fields_})
        try:
            _check(rc)
            outs = []
            for c in range(n):
                a = np.ctypeslib.as_array(ctypes.cast(coefs[c].fdata, ctypes.POINTER(ctypes.c_float)), shape=(coefs[c].h, coefs[c].w))
                outs.append(a.copy())
        finally:
            for c in range(n):
                libc.free(coefs[c].fdata)              # free_simd, jpeg2png.c:169
                libc.free(coefs[c].data)
    return outs, seconds


def compute(planes, weight, pweight, iterations, log=False, device=0):
    """Python twin of the reference's compute() (compute.h:8): same arguments and the same
    in/out convention — on return every plane's `fdata` is the W x H canvas plane and its
    w/h are rewritten to the canvas size (compute.c:455-461).  Returns the log rows when asked."""
    for p in planes:
        if p.fdata is None:
            raise J2PError("compute() expects decoded planes in fdata (jpeg.c:83-92); use decode_plane()")
    with Solver(planes, weight, pweight, iterations, device=device) as s:
        rows = s.run(iterations, log=log)
        outs = [s.download(c) for c in range(len(planes))]
        W, H = s.W, s.H
    for p, o in zip(planes, outs):
        p.fdata = o
        p.w, p.h = W, H
    return rows
