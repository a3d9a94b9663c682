"""instruction count of k_gradient's marches per row trip, from the device assembly — the kernel is issue-bound,
so this is the number to drive down.  One line per compiled march (in the order the compiler laid them out:
interior strips with unit sampling, interior strips, general) of the 1-channel and the channel-per-wavefront
kernel; blocks of the IEEE fallback (they contain v_div_scale / v_sqrt) are left out, the 16-row partial flush
(ds_bpermute) is counted although it runs on one trip in sixteen.
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


def count(body, trips):
    blocks, cur = [], []
    for line in body:
        t = line.strip()
        if (re.match(r"^\.LBB\d+_\d+:", t) or t.startswith("; %bb.")) and cur:      # labelled and fall-through blocks
            blocks.append(cur)
            cur = []
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
    vmem = sum(n for k, n in c.items() if k.startswith("global"))
    lds = sum(n for k, n in c.items() if k.startswith("ds_"))
    return (f"total {tot / trips:.1f} = VALU {valu / trips:.1f} + SALU {salu / trips:.1f} + s_nop {c['s_nop'] / trips:.1f} "
            f"+ s_waitcnt {c['s_waitcnt'] / trips:.1f} + vmem {vmem / trips:.1f} + lds {lds / trips:.1f}")


for kern, trips in (("_ZN3j2p10k_gradientILi1ELb1ELb0ELi1ELb1EEEvNS_8GradArgsE", 4), ("_ZN3j2p10k_gradientILi1ELb1ELb0ELi3ELb0EEEvNS_8GradArgsE", 3)):
    i = text.index(kern + ":")
    j = text.index("s_endpgm", i)
    L = text[i:j].split("\n")
    hdr = [n for n, line in enumerate(L) if "Inner Loop Header" in line]
    m = re.search(r"; NumVgprs: *(\d+)", text[j:j + 6000])
    code = re.search(r"codeLenInByte = (\d+)", text[j:j + 6000])
    print(f"{kern[-34:-16]}: VGPRs {m.group(1) if m else '?'}, code {code.group(1) if code else '?'} bytes")
    for which, h in enumerate(hdr):
        body = L[h:hdr[which + 1]] if which + 1 < len(hdr) else L[h:]
        print(f"   march {which}: per trip {count(body, trips)}")
