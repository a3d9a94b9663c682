#!/usr/bin/env python3
"""VGPR / SGPR / LDS / scratch use of every kernel, read from the device assembly's amdhsa metadata.
usage: python tools/kernel_resources.py [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jpeg2png_amd.buildlib import HIP_FLAGS, INCLUDE, CSRC  # noqa: E402

out = "/tmp/j2p_res.s"
cmd = ["/opt/rocm/bin/hipcc", *[f for f in HIP_FLAGS if f != "-Wall"], *sys.argv[1:], "-I", INCLUDE, "-I", CSRC,
       "--cuda-device-only", "-S", "-o", out, os.path.join(CSRC, "j2p_solver.hip")]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
md = open(out).read()
md = md[md.index("amdhsa.kernels"):]
for b in md.split("  - .agpr_count")[1:]:
    f = {k: re.search(r"\.%s:\s+(\S+)" % k, b).group(1) for k in
         ("name", "vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size")}
    name = subprocess.run(["c++filt", f["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"^void j2p::", "", name)
    print(f"{name[:70]:70s} vgpr {f['vgpr_count']:>4} sgpr {f['sgpr_count']:>4} "
          f"lds {f['group_segment_fixed_size']:>6} scratch {f['private_segment_fixed_size']}")
