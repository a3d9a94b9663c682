"""instruction count AND estimated issue cycles of k_gradient's marches per row trip, from the device assembly.
One line per compiled march (in the order the compiler laid them out — at present general, interior strips, interior
strips with unit sampling = the hot one, whose count runs on into the strip's epilogue) of the 1-channel (non-temporal g: the 4096^2 headline) and the channel-per-wavefront kernel; blocks
of the IEEE fallback (they contain v_div_scale / v_sqrt) are left out, the partial flush is counted although it runs
once per strip.  Cycle weights: tools/ubench/valu_rates (profiles/r03_valu_rates.json, 8 wavefronts per SIMD):
plain f32 2.7, packed f32 4.5-5.6, transcendental 8.1, f64 4.5-5.3, DPP 4.1, max / min / compare / select 4.3.
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


def cycles(op):
    if op.startswith("v_pk_fma"):
        return 5.6
    if op.startswith("v_pk_"):
        return 4.7
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log")):
        return 8.1
    if "f64" in op:
        return 5.0
    if op.endswith("_dpp"):
        return 4.1
    if op.startswith(("v_max", "v_min", "v_cmp", "v_cndmask")):
        return 4.3
    if op.startswith("v_"):
        return 2.7
    return 0.0


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
    pk = sum(n for k, n in c.items() if k.startswith("v_pk_"))
    dpp = sum(n for k, n in c.items() if k.endswith("_dpp"))
    trans = sum(n for k, n in c.items() if k.startswith(("v_rcp", "v_rsq", "v_sqrt")))
    salu = sum(n for k, n in c.items() if k.startswith("s_") and k not in ("s_nop", "s_waitcnt"))
    vmem = sum(n for k, n in c.items() if k.startswith(("global", "buffer_load", "buffer_store")))
    lds = sum(n for k, n in c.items() if k.startswith("ds_"))
    cyc = sum(n * cycles(k) for k, n in c.items())
    return (f"total {tot / trips:.1f} = VALU {valu / trips:.1f} (packed {pk / trips:.1f}, DPP {dpp / trips:.1f}, transcendental {trans / trips:.1f}) "
            f"+ SALU {salu / trips:.1f} + s_nop {c['s_nop'] / trips:.1f} + s_waitcnt {c['s_waitcnt'] / trips:.1f} + vmem {vmem / trips:.1f} "
            f"+ lds {lds / trips:.1f};  VALU issue ~{cyc / trips:.0f} cycles")


names = {}
for m in re.finditer(r"^(_ZN3j2p10k_gradient\w+):", text, re.M):
    dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    names[dem] = m.group(1)
for want, trips in (("k_gradient<1, true, false, 1, 1, 2>", 4), ("k_gradient<1, true, false, 1, 0, 1>", 4), ("k_gradient<1, true, false, 3, 0, 2>", 4),
                    ("k_gradient<1, true, false, 3, 0, 1>", 4)):
    kern = next((v for k, v in names.items() if want in k), None)
    if not kern:
        continue
    i = text.index(kern + ":")
    j = text.index("s_endpgm", i)
    L = text[i:j].split("\n")
    # marches are depth-1 loops; one that wraps a store in a waterfall loop (a row offset the compiler kept in a vector
    # register) is a depth-1 header with inner loops, not an "Inner Loop Header": take both
    hdr = [n for n, line in enumerate(L) if "Loop Header: Depth=1" in line]
    m = re.search(r"; NumVgprs: *(\d+)", text[j:j + 6000])
    code = re.search(r"codeLenInByte = (\d+)", text[j:j + 6000])
    print(f"{want}: VGPRs {m.group(1) if m else '?'}, code {code.group(1) if code else '?'} bytes")
    for which, h in enumerate(hdr):
        body = L[h:hdr[which + 1]] if which + 1 < len(hdr) else L[h:]
        if sum(1 for t in body if t.strip().startswith("v_pk_")) < 100:
            continue                        # not a march (reduction / flush loops)
        print(f"   march {which}: per trip {count(body, trips)}")
