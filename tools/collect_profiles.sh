#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the rocprofv3 evidence bench.py's roofline object refers to.
#   gpurun -- 'bash tools/collect_profiles.sh r01'
# leaves under gpurun_out/<tag>_*: the kernel-trace stats of the default bench command, the bench JSON printed
# under the profiler, and one counter pass each for FETCH_SIZE, WRITE_SIZE and the SQ/GRBM set (PMC passes are
# separate runs with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-host-to-host"
B2="python $R/bench.py --steps 1 --warmup 0 --iterations 50 --no-cpu-baseline --no-other-configs --no-host-to-host"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats" -- $B > "$OUT/${TAG}_bench_under_rocprof.log" 2>&1
grep '^{' "$OUT/${TAG}_bench_under_rocprof.log" | tail -1 > "$OUT/${TAG}_bench_under_rocprof.json"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_pmc_fetch" -- $B2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_pmc_write" -- $B2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE \
        --output-format csv -d "$OUT/${TAG}_pmc_sq" -- $B2 > /dev/null 2>&1
# keep what is small: the stats csv and the per-kernel counter averages
find "$OUT/${TAG}_stats" -name '*kernel_stats.csv' -exec cp {} "$OUT/${TAG}_bench_kernel_stats.csv" \;
python $R/tools/pmc_summary.py --about "rocprofv3 on MI355X, $TAG: separate --kernel-trace --pmc passes (FETCH_SIZE | WRITE_SIZE | the SQ/GRBM set) of \`python bench.py --steps 1 --warmup 0 --iterations 50 --no-cpu-baseline --no-other-configs --no-host-to-host\` (4096x4096 Y-only Q10, BASELINE configs[2]), averaged over the launches of each kernel; hbm_* = FETCH_SIZE x2 (gfx950 correction of MI355X_MICROARCH.md) and WRITE_SIZE as reported. Collected by tools/collect_profiles.sh." \
        "$OUT/${TAG}_pmc_fetch" "$OUT/${TAG}_pmc_write" "$OUT/${TAG}_pmc_sq" > "$OUT/${TAG}_pmc_summary.json"
find "$OUT" -name '*.csv' -size +2M -delete
find "$OUT" -name '*_agent_info.csv' -delete
python $R/bench.py --steps 3 --warmup 1 > "$OUT/${TAG}_bench_n1.log" 2>&1
grep '^{' "$OUT/${TAG}_bench_n1.log" | tail -1 > "$OUT/${TAG}_bench_n1.json"
cat "$OUT/${TAG}_bench_kernel_stats.csv" | head -8
cat "$OUT/${TAG}_bench_n1.json"
