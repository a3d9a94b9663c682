#!/usr/bin/env python3
"""Build another copy of the library with extra compiler flags, for same-box A/B timing of kernel variants:
    python tools/build_variant.py NAME -DJ2P_EXP_...      ->  ab/libj2p_NAME.so
    J2P_LIBRARY=ab/libj2p_NAME.so python bench.py ...
(ab/ is git-ignored but travels to the GPU box with gpurun: keep only the builds the next call needs and delete them
afterwards — every file here is pushed with every lease.  The object file is removed once the library is linked.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jpeg2png_amd.buildlib import CSRC, HIP_FLAGS, HIP_UNITS, INCLUDE, build  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
build()                                     # the host-only objects are shared with the normal build
out_dir = os.path.join(ROOT, "ab")
os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, f"j2p_solver_{name}.o")
subprocess.run(["/opt/rocm/bin/hipcc", *HIP_FLAGS, *flags, "-I", INCLUDE, "-I", CSRC, "-c", os.path.join(CSRC, "j2p_solver.hip"), "-o", obj], check=True)
objs = [obj] + [os.path.join(CSRC, u.replace(".hip", ".o")) for u in HIP_UNITS if u != "j2p_solver.hip"] + [os.path.join(CSRC, "compute_host.o")]
lib = os.path.join(out_dir, f"libj2p_{name}.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-lpthread", "-o", lib], check=True)
os.remove(obj)
print(lib)
