#!/bin/bash
# round 3, call E: kernel trace of a band that has the GPU to itself (where do its 45 us per iteration over the
# whole-canvas solve go?)
set -u
O=$(pwd)/gpurun_out/r03e
R=$(pwd)
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for side in 0 1; do
  J2P_TILED_SIDE=$side rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_side$side -- python $R/tools/band_alone.py > $O/band_alone_side$side.log 2>&1
  tail -2 $O/band_alone_side$side.log
done
cd $R
python - <<'PY'
import csv, glob, collections, json
for side in (0, 1):
    f = glob.glob(f"gpurun_out/r03e/trace_side{side}/**/*kernel_trace.csv", recursive=True)
    if not f:
        print("no trace", side); continue
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the tiled run comes first (3+1 solves of 100 iterations), then the whole-canvas solver: split at the first k_norm_whole
    names = [r["Kernel_Name"] for r in rows]
    cut = next((i for i, n in enumerate(names) if "k_norm_whole" in n), len(rows))
    tiled = rows[:cut]
    # last 40 % of the tiled part: steady state
    seg = tiled[int(len(tiled) * 0.6):]
    def short(n):
        n = n.replace("void j2p::", "").split("(")[0]
        return n[:60]
    dur = collections.defaultdict(list)
    for r in seg:
        dur[short(r["Kernel_Name"]) + " grid=" + r.get("Grid_Size_X", r.get("Grid_Size", "?")) + "x" + r.get("Grid_Size_Y", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    nit = sum(1 for r in seg if "k_norm_bands" in r["Kernel_Name"]) / 2.0        # two bands
    print(f"side={side}: steady-state window {(t1 - t0) / 1e3:.0f} us, ~{nit:.0f} iterations, {(t1 - t0) / 1e3 / max(nit, 1):.1f} us per iteration")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f"   {k:95s} n={len(v):5d} mean {sum(v) / len(v):8.2f} us  sum/iter {sum(v) / max(nit, 1):8.2f}")
    # busy union of band 0's big kernels vs window: idle time
    ivs = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
    busy, cur_s, cur_e = 0, ivs[0][0], ivs[0][1]
    for a, b in ivs[1:]:
        if a > cur_e:
            busy += cur_e - cur_s; cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    busy += cur_e - cur_s
    print(f"   GPU has a kernel running {busy / (t1 - t0) * 100:.1f} % of the window; idle {(t1 - t0 - busy) / 1e3 / max(nit, 1):.1f} us per iteration")
PY
find gpurun_out/r03e -name '*.csv' -size +1M -delete; find gpurun_out/r03e -name '*agent_info*' -delete
