"""instruction count of k_gradient's common path (interior strips, screened arithmetic) per row trip, from the
device assembly — the kernel is issue-bound, so this is the number to drive down.
usage: python tools/isa_count.py [extra hipcc flags...]   (compiles jpeg2png_amd/csrc/j2p_solver.hip to /tmp)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jpeg2png_amd.buildlib import HIP_FLAGS, INCLUDE, CSRC

out = "/tmp/j2p_isa.s"
cmd = ["/opt/rocm/bin/hipcc", *[f for f in HIP_FLAGS if f not in ("-Wall",)], *sys.argv[1:], "-I", INCLUDE, "-I", CSRC,
       "--cuda-device-only", "-S", "-o", out, os.path.join(CSRC, "j2p_solver.hip")]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
text = open(out).read()
for kern, trips in (("_ZN3j2p10k_gradientILi1ELb1ELb0ELi1EEEvNS_8GradArgsE", 4), ("_ZN3j2p10k_gradientILi1ELb1ELb0ELi3EEEvNS_8GradArgsE", 3)):
    i = text.index(kern + ":")
    j = text.index("s_endpgm", i)
    L = text[i:j].split("\n")
    hdr = [n for n, l in enumerate(L) if "Inner Loop Header" in l]
    body = L[hdr[-1]:]                      # the second loop = the interior-strip (FREE) march
    # basic blocks; drop those of the IEEE fallback (they contain v_div_scale / v_sqrt)
    blocks, cur = [], []
    for l in body:
        t = l.strip()
        if (re.match(r"^\.LBB\d+_\d+:", t) or t.startswith("; %bb.")) and cur:      # labelled and fall-through blocks
            blocks.append(cur); cur = []
        cur.append(t)
    blocks.append(cur)
    c = collections.Counter()
    for b in blocks:
        if any(x.startswith(("v_div_scale", "v_sqrt_f32")) for x in b):
            continue
        for t in b:
            if not t or t[0] in ".;" or t.endswith(":"):
                continue
            c[t.split()[0]] += 1
    tot = sum(c.values())
    valu = sum(n for k, n in c.items() if k.startswith("v_"))
    salu = sum(n for k, n in c.items() if k.startswith("s_") and k not in ("s_nop", "s_waitcnt"))
    m = re.search(r"; NumVgprs: *(\d+)", text[j:j + 6000])
    print(f"{kern[-28:-16]}: per trip total {tot / trips:.1f} = VALU {valu / trips:.1f} + SALU {salu / trips:.1f} + s_nop {c['s_nop'] / trips:.1f} "
          f"+ s_waitcnt {c['s_waitcnt'] / trips:.1f} + vmem {sum(n for k, n in c.items() if k.startswith('global')) / trips:.1f} "
          f"+ lds {sum(n for k, n in c.items() if k.startswith('ds_')) / trips:.1f}; VGPRs {m.group(1) if m else '?'}")
