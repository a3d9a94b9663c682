"""Build libjpeg2png_amd.so (HIP kernels + C-ABI shim + C host drop-in) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only container too.
Device code is built with -ffp-contract=off and no fast-math: the solver has to
reproduce the reference arithmetic operation for operation.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(HERE, "libjpeg2png_amd.so")

HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off",            # no fused multiply-add: reference is built with -ffp-contract=off (Makefile:41-45)
    "-fno-fast-math",
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-fno-gpu-flush-denormals-to-zero",
    # the kernels say themselves which float operations are packed (float2 arithmetic); left to the SLP pass, pairs of scalar
    # operations — the DCT butterflies of k_project above all — are packed too, with v_mov / v_pk_mov to line the operands
    # up, and a packed f32 operation costs two plain ones on gfx950 anyway (profiles/r03_valu_rates.json).  Same bits;
    # measured 4096^2 136.8 -> 140.6 Gpx-it/s (k_project 65.4 -> 62.0 us), 16384x2048 141.9 -> 146.3
    "-fno-slp-vectorize",
    "-Wall", "-Wno-unused-function",
]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


HIP_UNITS = ("j2p_solver.hip", "j2p_tiled.hip", "j2p_batch.hip")
HEADERS = ("j2p_kernels.hip.h", "j2p_internal.h")


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("build failed: " + " ".join(cmd))
    if verbose and (r.stdout or r.stderr):
        print(r.stdout + r.stderr)


def build(force=False, verbose=False):
    units = [u for u in HIP_UNITS if os.path.exists(os.path.join(CSRC, u))]
    common = [os.path.join(CSRC, f) for f in HEADERS]
    common += [os.path.join(INCLUDE, f) for f in ("jpeg2png_amd.h", "jpeg2png_amd_compute.h")]
    common.append(os.path.abspath(__file__))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    gcc = shutil.which("gcc") or "gcc"
    objs = []
    extra = os.environ.get("J2P_CXXFLAGS", "").split()
    jobs = []
    for u in units:
        src, obj = os.path.join(CSRC, u), os.path.join(CSRC, u.replace(".hip", ".o"))
        objs.append(obj)
        # only j2p_solver.hip holds device code (it includes the kernels header); the others are host code
        deps = [src] + (common if u == "j2p_solver.hip" else common[1:])
        if force or _newer(obj, deps):
            jobs.append([hipcc, *HIP_FLAGS, *extra, "-I", INCLUDE, "-I", CSRC, "-c", src, "-o", obj])
    csrc, cobj = os.path.join(CSRC, "compute_host.c"), os.path.join(CSRC, "compute_host.o")
    objs.append(cobj)
    if force or _newer(cobj, [csrc] + common[1:]):
        jobs.append([gcc, "-std=c11", "-O2", "-fPIC", "-Wall", "-Wextra", "-ffp-contract=off", "-I", INCLUDE, "-c", csrc, "-o", cobj])
    if not jobs and not _newer(LIB, objs):
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(lambda c: _run(c, verbose), jobs))
    # SONAME so that programs linked with -ljpeg2png_amd find the library through their RUNPATH ($ORIGIN),
    # wherever the checkout lives
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-lpthread", "-Wl,-soname,libjpeg2png_amd.so", "-o", LIB], verbose)
    return LIB


DEBUG_LIB = os.path.join(HERE, "libjpeg2png_amd_debug.so")


def build_debug(force=False, verbose=False):
    """The checked build (-DJ2P_DEBUG): every global access of the two phase kernels is compared with the range
    it is meant to stay in, the counterpart of the reference's DEBUG=1 build with its asserting pixel indexer
    (utils.h:68-81).  Same sources; only the device translation unit is compiled a second time.  Loaded with
    J2P_LIBRARY=<this file> (tests/test_debug_build_gpu.py, tools/debug_sweep.py)."""
    build(force=False, verbose=verbose)             # the host-only objects are shared with the release build
    src = os.path.join(CSRC, "j2p_solver.hip")
    obj = os.path.join(CSRC, "j2p_solver_debug.o")
    deps = [src] + [os.path.join(CSRC, f) for f in HEADERS] + [os.path.join(INCLUDE, "jpeg2png_amd.h"), os.path.abspath(__file__)]
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if force or _newer(obj, deps):
        _run([hipcc, *HIP_FLAGS, "-DJ2P_DEBUG", "-I", INCLUDE, "-I", CSRC, "-c", src, "-o", obj], verbose)
    objs = [obj] + [os.path.join(CSRC, u.replace(".hip", ".o")) for u in HIP_UNITS if u != "j2p_solver.hip"]
    objs.append(os.path.join(CSRC, "compute_host.o"))
    if force or _newer(DEBUG_LIB, objs):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-lpthread", "-Wl,-soname,libjpeg2png_amd_debug.so", "-o", DEBUG_LIB], verbose)
    return DEBUG_LIB


EXP_LIB = os.path.join(HERE, "libjpeg2png_amd_exp.so")


def build_experiments(force=False, verbose=False):
    """The experiments build (-DJ2P_EXPERIMENTS, jpeg2png_amd/libjpeg2png_amd_exp.so): the release sources plus the
    schedules that lost their measurements (one column per lane, all channels of a joint image in one wavefront, split
    phases) and the environment knobs that select them or move the shares of a gradient launch's item sizes
    (j2p_internal.h: j2p_exp_env).  What the schedule-equivalence tests load (conftest.exp_lib)
    and the timing tools run on (J2P_LIBRARY=<this file>); never what a user of the library gets."""
    build(force=False, verbose=verbose)             # compute_host.o is shared
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    common = [os.path.join(CSRC, f) for f in HEADERS] + [os.path.join(INCLUDE, "jpeg2png_amd.h"), os.path.abspath(__file__)]
    # (the objects do not travel with a gpurun lease, the library does: up to date against the SOURCES is enough)
    sources = [os.path.join(CSRC, u) for u in HIP_UNITS] + common + [os.path.join(CSRC, "compute_host.c")]
    if not force and os.path.exists(EXP_LIB) and not _newer(EXP_LIB, sources):
        return EXP_LIB
    jobs, objs = [], []
    for u in HIP_UNITS:
        src, obj = os.path.join(CSRC, u), os.path.join(CSRC, u.replace(".hip", "_exp.o"))
        objs.append(obj)
        deps = [src] + (common if u == "j2p_solver.hip" else common[1:])
        if force or _newer(obj, deps):
            jobs.append([hipcc, *HIP_FLAGS, "-DJ2P_EXPERIMENTS", "-I", INCLUDE, "-I", CSRC, "-c", src, "-o", obj])
    objs.append(os.path.join(CSRC, "compute_host.o"))
    if jobs or _newer(EXP_LIB, objs):
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(lambda c: _run(c, verbose), jobs))
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-lpthread", "-Wl,-soname,libjpeg2png_amd_exp.so", "-o", EXP_LIB], verbose)
    return EXP_LIB


CLI = os.path.join(HERE, "jpeg2png_gpu")


def build_cli(force=False, verbose=False, prefix=None):
    """Build the command-line driver cli/jpeg2png_gpu.c (needs libjpeg + libpng headers; this image
    ships them under /opt/conda).  Returns the path, or None when the libraries are not available."""
    prefix = prefix or os.environ.get("J2P_IMG_PREFIX", "/opt/conda")
    src = os.path.join(ROOT, "cli", "jpeg2png_gpu.c")
    if not (os.path.exists(os.path.join(prefix, "include", "jpeglib.h")) and os.path.exists(os.path.join(prefix, "include", "png.h"))):
        return None
    lib = build(force=False, verbose=verbose)
    if not force and not _newer(CLI, [src, lib, os.path.join(INCLUDE, "jpeg2png_amd.h")]):
        return CLI
    gcc = shutil.which("gcc") or "gcc"
    plib = os.path.join(prefix, "lib")
    # the image libraries are named by full path and found at run time through RUNPATH (direct
    # dependencies only), so that the HIP runtime keeps resolving libstdc++ from the system
    cmd = [gcc, "-std=c11", "-O2", "-Wall", "-Wextra", "-I", INCLUDE, "-I", os.path.join(prefix, "include"), src, "-o", CLI,
           "-L", HERE, "-ljpeg2png_amd", os.path.join(plib, "libjpeg.so"), os.path.join(plib, "libpng16.so"), os.path.join(plib, "libz.so"), "-lpthread",
           "-Wl,--enable-new-dtags", "-Wl,-rpath-link,/usr/lib/x86_64-linux-gnu:/opt/rocm/lib",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + plib]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("CLI build failed")
    return CLI


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_cli(force="--force" in sys.argv, verbose=True))
