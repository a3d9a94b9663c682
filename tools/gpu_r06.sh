#!/bin/bash
# Round 6's GPU calls, one section per call (bash tools/gpu_r06.sh <section>); everything lands under gpurun_out/r06_*.
set -u
S=${1:-a}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
BENCH="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-host-to-host"
line() { grep '^{' | tail -1; }
brief() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['per_kernel']
print(json.dumps({'variant': '$1', 'Mpx_it_per_s': d['value'] or d.get('unverified_value'), 'us_per_iteration': round(r['iteration_ms']*1e3,2), 'k_gradient_us': round(k['k_gradient']['avg_launch_ms']*1e3,1), 'k_project_us': round(k['k_project']['avg_launch_ms']*1e3,1), 'bit_identical_to_reference': (d.get('parity') or {}).get('bit_identical')}))"; }
sized() {  # sized W H ITER VARIANT [LIB]
  ( J2P_LIBRARY=${5:-} timeout 300 python bench.py --size $1 --height $2 --iterations $3 --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-to-host ) 2>/dev/null | line | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'plane': '$1x$2', 'variant': '$4', 'us_per_iteration': round(r['iteration_ms']*1e3,2), 'frac': r['frac'], 'k_gradient_us': round(r['per_kernel']['k_gradient']['avg_launch_ms']*1e3,1), 'k_project_us': round(r['per_kernel']['k_project']['avg_launch_ms']*1e3,1)}))"
}
case $S in
a)
  # the per-SIMD picture of the final round-5 kernels (review item 1), then rows per strip at 4096^2 (timing only: J2P_RPW
  # other than 16 changes the order of the norm's partial sums)
  ( timeout 300 python __graft_entry__.py --smoke ) 2>&1 | tail -1
  J2P_LIBRARY=ab/libj2p_trace.so timeout 300 python tools/wave_trace.py 4096 4096 444 y 12 > $O/r06_wave_trace.jsonl 2>$O/r06_wave_trace.err; tail -c 2500 $O/r06_wave_trace.jsonl
  for rep in 1 2; do
    for rpw in 16 17 18 20 24 32 33 34 36 40 48; do
      J2P_RPW=$rpw sized 4096 4096 100 rpw$rpw jpeg2png_amd/libjpeg2png_amd_exp.so
    done
  done | tee $O/r06_rpw_4096.jsonl
  ;;
b)
  # half / quarter items at the end of the gradient launch: parity first, then the shares at 4096^2 (timing; bits are equal by test)
  ( timeout 900 python -m pytest tests/test_parity_gpu.py -q -x --timeout 600 -k "half_and_quarter or schedule_switch or joint_modes or full_size_against or band" ) > $O/r06_b_tests.log 2>&1; echo "tests rc=$?"; tail -8 $O/r06_b_tests.log
  for zb in 0 13 26 38 51 64; do
    for zc in 0 5 10 15 20 26; do
      J2P_ZONE_B=$zb J2P_ZONE_C=$zc sized 4096 4096 100 zones_${zb}_${zc} jpeg2png_amd/libjpeg2png_amd_exp.so
    done
  done | tee $O/r06_zones_4096.jsonl
  ;;
c)
  # the per-SIMD picture with half / quarter items at the end of the launch
  for z in "0 0" "26 10" "51 26" "100 50"; do
    set -- $z
    J2P_ZONE_B=$1 J2P_ZONE_C=$2 J2P_LIBRARY=ab/libj2p_tracex.so timeout 300 python tools/wave_trace.py 4096 4096 444 y 12 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())['launches']['k_gradient']; p=d['per_simd']
print(json.dumps({'zones': '$1/$2', 'wavefronts': d['wavefronts'], 'span_us': p['span_us'], 'life': d['wave_life_us_p10_p50_p90_max'], 'busy_until': p['busy_until_us_p10_p50_p90_max'], 'mean_resident': p['mean_resident_wavefronts_per_simd'], 'resident_every_5us': p['resident_wavefronts_per_simd_over_time_every_5us'], 'start_p50_p90_max': d['wave_start_us_p50_p90_max']}))"
  done | tee $O/r06_wave_trace_zones.jsonl
  ;;
d)
  # who are the long-lived wavefronts of k_gradient (no zones)?
  J2P_LIBRARY=ab/libj2p_tracex.so J2P_ZONE_B=0 J2P_ZONE_C=0 timeout 300 python tools/wave_trace.py 4096 4096 444 y 12 2>$O/r06_d.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())['launches']['k_gradient']['who_lives_long']
for k,v in d.items(): print(k, json.dumps(v))" | tee $O/r06_who_lives_long.txt
  tail -3 $O/r06_d.err
  ;;
e)
  # workgroups of four consecutive (tile row, strip) pairs: parity, the per-SIMD picture, then zones by canvas size
  ( timeout 900 python -m pytest tests/test_parity_gpu.py -q -x --timeout 600 -k "half_and_quarter or schedule_switch or joint_modes or full_size_against or band" ) > $O/r06_e_tests.log 2>&1; echo "tests rc=$?"; tail -8 $O/r06_e_tests.log
  for z in "0 0" "32 10"; do
    set -- $z
    J2P_ZONE_B=$1 J2P_ZONE_C=$2 J2P_LIBRARY=ab/libj2p_tracex.so timeout 300 python tools/wave_trace.py 4096 4096 444 y 12 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())['launches']['k_gradient']; p=d['per_simd']
print(json.dumps({'zones': '$1/$2', 'wavefronts': d['wavefronts'], 'span_us': p['span_us'], 'life': d['wave_life_us_p10_p50_p90_max'], 'per_simd_histogram': p['wavefronts_per_simd_histogram'], 'busy_until': p['busy_until_us_p10_p50_p90_max'], 'mean_resident': p['mean_resident_wavefronts_per_simd'], 'resident_every_5us': p['resident_wavefronts_per_simd_over_time_every_5us'], 'start_p50_p90_max': d['wave_start_us_p50_p90_max']}))"
  done | tee $O/r06_wave_trace_linear.jsonl
  for sz in "2048 2048" "4096 2048" "4096 4096" "8192 4096" "16384 2048" "8192 8192"; do
    set -- $sz
    for z in "0 0" "32 10" "64 20" "0 0" "32 10"; do
      set -- $sz $z
      J2P_ZONE_B=$3 J2P_ZONE_C=$4 sized $1 $2 100 zones_$3_$4 jpeg2png_amd/libjpeg2png_amd_exp.so
    done
  done | tee $O/r06_zones_by_size.jsonl
  ;;
f)
  # half / quarter shares on mid-size canvases (1080p: 8-row tile rows, halves only)
  ( timeout 600 python -m pytest tests/test_parity_gpu.py -q -x --timeout 600 -k "half_and_quarter or drop_in" ) > $O/r06_f_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/r06_f_tests.log
  for sz in "1920 1080" "2048 2048" "3072 2048" "4096 2048" "4096 3072" "4096 4096"; do
    for z in "0 0" "32 10" "64 20" "96 32" "128 40" "160 64" "256 0" "0 0"; do
      set -- $sz $z
      J2P_ZONE_B=$3 J2P_ZONE_C=$4 sized $1 $2 100 zones_$3_$4 jpeg2png_amd/libjpeg2png_amd_exp.so
    done
  done | tee $O/r06_zones_mid_sizes.jsonl
  ;;
g)
  # canvases past the Infinity Cache (review item 2): real bytes at the L2's memory side, L2 hit rates, address translation;
  # then the knobs one at a time at 8192^2 and 16384x2048: rows per strip 32 (timing only), ring of five row slots
  cd /tmp
  rocprofv3 --list-avail 2>/dev/null | grep -i -E "UTCL|TLB|TCC_HIT|TCC_MISS|TCC_EA0_RDREQ|TCC_EA0_WRREQ|TCP_TCC_READ|TCC_REQ|MALL|TCC_BUBBLE|TCC_EA0_RD_UNCACHED" | cut -c1-160 | sort -u | head -60 > $GRAFT_REPO_ROOT/$O/r06_counters_available.txt
  cd $GRAFT_REPO_ROOT
  for sz in "4096 4096" "8192 4096" "8192 8192" "16384 4096" "16384 8192"; do
    set -- $sz
    B2="python $GRAFT_REPO_ROOT/bench.py --size $1 --height $2 --steps 1 --warmup 0 --iterations 20 --no-cpu-baseline --no-other-configs --no-host-to-host"
    T=$O/r06_pmc_$1x$2
    ( cd /tmp
      rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/${T}_stats -- $B2 > /dev/null 2>&1
      rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/${T}_fetch -- $B2 > /dev/null 2>&1
      rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/${T}_write -- $B2 > /dev/null 2>&1
      rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $GRAFT_REPO_ROOT/${T}_l2 -- $B2 > /dev/null 2>&1
      rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum --output-format csv -d $GRAFT_REPO_ROOT/${T}_tlb -- $B2 > /dev/null 2>&1 )
    find ${T}_stats -name '*kernel_stats.csv' -exec cp {} ${T}_kernel_stats.csv \;
    python tools/pmc_summary.py --about "$1x$2 Y, -i 20" ${T}_fetch ${T}_write ${T}_l2 ${T}_tlb > ${T}.json
    rm -rf ${T}_stats ${T}_fetch ${T}_write ${T}_l2 ${T}_tlb
    python -c "
import json; d=json.load(open('${T}.json'))
for k in ('j2p::k_gradient','j2p::k_project'):
    v=[x for n,x in d.items() if n.startswith(k)]
    print('$1x$2', k, json.dumps(v[0] if v else None))"
    head -4 ${T}_kernel_stats.csv | cut -c1-200
  done 2>&1 | tee $O/r06_pmc_big.log
  for sz in "8192 8192" "16384 2048"; do
    set -- $sz
    for v in base rpw32 base ring5 rpw24; do
      case $v in
        base) sized $1 $2 50 base jpeg2png_amd/libjpeg2png_amd_exp.so ;;
        rpw32) J2P_RPW=32 sized $1 $2 50 rpw32 jpeg2png_amd/libjpeg2png_amd_exp.so ;;
        rpw24) J2P_RPW=24 sized $1 $2 50 rpw24 jpeg2png_amd/libjpeg2png_amd_exp.so ;;
        ring5) sized $1 $2 50 ring5 ab/libj2p_bigring5.so ;;
esac
    done
  done | tee $O/r06_big_knobs.jsonl
  ;;
h)
  # does the row stride matter (a power-of-two stride keeps a strip's rows on one set of HBM channels)?  Same pixels per row +- 64
  for sz in "4096 4096" "4160 4096" "8192 4096" "8256 4096" "8192 8192" "8256 8192" "8320 8192" "16384 2048" "16448 2048" "16384 4096" "16448 4096" "12288 8192" "8192 8192" "8256 8192"; do
    set -- $sz
    sized $1 $2 50 stride_probe
  done | tee $O/r06_stride_probe.jsonl
  ;;
i)
  # the gradient phase bottom-up, the projection top-down: each phase starts on what the Infinity Cache still holds
  ( timeout 600 python -m pytest tests/test_parity_gpu.py -q -x --timeout 600 -k "half_and_quarter or schedule_switch" ) > $O/r06_i_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r06_i_tests.log
  for sz in "4096 4096" "8192 4096" "16384 2048" "8192 8192" "16384 4096" "16384 8192"; do
    set -- $sz
    for v in 0 1 0 1; do
      J2P_GRAD_REVERSE=$v sized $1 $2 50 reverse$v jpeg2png_amd/libjpeg2png_amd_exp.so
    done
  done | tee $O/r06_reverse.jsonl
  ;;
j)
  # the whole GPU suite on the round's library so far
  ( timeout 1700 python -m pytest tests -m gpu -q --durations=8 --timeout 900 ) > $O/r06_j_suite.log 2>&1; echo "suite rc=$?"; tail -40 $O/r06_j_suite.log
  ;;
k)
  # one band's counters (review item 3a), the batch engine's occupancy (item 4), the 2-rank launch shape on one GPU (item 3)
  T=$O/r06_pmc_band
  ( cd /tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/${T}_fetch -- python $GRAFT_REPO_ROOT/tools/band_pmc.py > /dev/null 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/${T}_write -- python $GRAFT_REPO_ROOT/tools/band_pmc.py > /dev/null 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/${T}_stats -- python $GRAFT_REPO_ROOT/tools/band_pmc.py > /dev/null 2>&1 )
  find ${T}_stats -name '*kernel_stats.csv' -exec cp {} ${T}_kernel_stats.csv \;
  python tools/pmc_summary.py --about "two 2048-row bands of the 16384-wide plane on one GPU through j2p_tiled, exchange direct, -i 20 (tools/band_pmc.py): per launch = per band" ${T}_fetch ${T}_write > ${T}.json
  python - <<PY
import json
d=json.load(open("${T}.json")); d["_shape"]=[16384, 2048]
json.dump(d, open("${T}.json","w"), indent=1)
print({k:v.get("hbm_bytes_per_launch") for k,v in d.items() if isinstance(v,dict) and "hbm_bytes_per_launch" in v})
PY
  rm -rf ${T}_fetch ${T}_write ${T}_stats; head -4 ${T}_kernel_stats.csv | cut -c1-220
  ( cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/r06_batch_trace -- python $GRAFT_REPO_ROOT/bench.py --config batch --steps 1 --warmup 1 --batch 32 > $GRAFT_REPO_ROOT/$O/r06_batch_trace.log 2>&1 )
  F=$(find $O/r06_batch_trace -name '*kernel_trace.csv' | head -1); python tools/batch_occupancy.py $F 64 | tee $O/r06_batch_occupancy.json; rm -rf $O/r06_batch_trace
  grep '^{' $O/r06_batch_trace.log | tail -1 | cut -c1-300
  ( J2P_BENCH_ONE_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --size 4096 ) > $O/r06_bench_2ranks.log 2>&1; echo "2-rank bench rc=$?"; grep '^{' $O/r06_bench_2ranks.log | tail -1 > $O/r06_bench_2ranks_1gpu_gloo.json; cut -c1-1500 $O/r06_bench_2ranks_1gpu_gloo.json; grep "^bench:" $O/r06_bench_2ranks.log | head -5
  ;;
l)
  # double items at the start of the launch: parity, then shares by size (timing)
  ( timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_batch_gpu.py -q -x --timeout 600 -k "half_and_quarter or schedule_switch or tile_gate or wrong_kind or band" ) > $O/r06_l_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/r06_l_tests.log
  for sz in "2048 2048" "4096 2048" "4096 4096" "16384 2048" "8192 8192"; do
    for z in "0 32 10" "64 32 10" "128 32 10" "176 32 10" "200 24 8" "230 16 6" "256 0 0" "0 32 10"; do
      set -- $sz $z
      J2P_ZONE_D=$3 J2P_ZONE_B=$4 J2P_ZONE_C=$5 sized $1 $2 100 zones_$3_$4_$5 jpeg2png_amd/libjpeg2png_amd_exp.so
    done
  done | tee $O/r06_doubles.jsonl
  ;;
m)
  # zig-zag (gradient bottom-up, projection top-down) x non-temporal level: does the alternation let MORE stay in the Infinity Cache?
  ntsized() {  # W H ITER NT REV
    ( J2P_GRAD_REVERSE=$5 J2P_LIBRARY=jpeg2png_amd/libjpeg2png_amd_exp.so timeout 300 python bench.py --size $1 --height $2 --iterations $3 --nt $4 --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-to-host ) 2>/dev/null | line | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'plane': '$1x$2', 'nt': $4, 'reverse': $5, 'us_per_iteration': round(r['iteration_ms']*1e3,2), 'frac': r['frac'], 'k_gradient_us': round(r['per_kernel']['k_gradient']['avg_launch_ms']*1e3,1), 'k_project_us': round(r['per_kernel']['k_project']['avg_launch_ms']*1e3,1)}))"
  }
  for sz in "4096 4096" "4096 5120" "8192 4096" "16384 2048" "8192 8192"; do
    set -- $sz
    for nt in 0 1 2 3; do
      for rev in 0 1; do ntsized $1 $2 100 $nt $rev; done
    done
  done | tee $O/r06_zigzag_nt.jsonl
  ;;
n)
  # counters again on the final kernels (the projection reads the coefficients as bytes now): per shape, then one band
  for sz in "4096 4096" "8192 4096" "8192 8192" "16384 4096" "16384 8192"; do
    set -- $sz
    B2="python $GRAFT_REPO_ROOT/bench.py --size $1 --height $2 --steps 1 --warmup 0 --iterations 20 --no-cpu-baseline --no-other-configs --no-host-to-host"
    T=$O/r06_pmc_$1x$2
    ( cd /tmp
      rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/${T}_stats -- $B2 > /dev/null 2>&1
      rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/${T}_fetch -- $B2 > /dev/null 2>&1
      rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/${T}_write -- $B2 > /dev/null 2>&1
      rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $GRAFT_REPO_ROOT/${T}_l2 -- $B2 > /dev/null 2>&1
      rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum --output-format csv -d $GRAFT_REPO_ROOT/${T}_tlb -- $B2 > /dev/null 2>&1 )
    find ${T}_stats -name '*kernel_stats.csv' -exec cp {} ${T}_kernel_stats.csv \;
    python tools/pmc_summary.py --about "$1x$2 Y, -i 20, final round-6 kernels (coefficients read as bytes)" ${T}_fetch ${T}_write ${T}_l2 ${T}_tlb > ${T}.json
    rm -rf ${T}_stats ${T}_fetch ${T}_write ${T}_l2 ${T}_tlb
    python -c "
import json; d=json.load(open('${T}.json'))
for k in ('j2p::k_gradient','j2p::k_project'):
    v=[x for n,x in d.items() if n.startswith(k)]
    print('$1x$2', k, json.dumps(v[0] if v else None)[:400])"
  done 2>&1 | tee $O/r06_pmc_big.log
  T=$O/r06_pmc_band
  ( cd /tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/${T}_fetch -- python $GRAFT_REPO_ROOT/tools/band_pmc.py > /dev/null 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/${T}_write -- python $GRAFT_REPO_ROOT/tools/band_pmc.py > /dev/null 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/${T}_stats -- python $GRAFT_REPO_ROOT/tools/band_pmc.py > /dev/null 2>&1 )
  find ${T}_stats -name '*kernel_stats.csv' -exec cp {} ${T}_kernel_stats.csv \;
  python tools/pmc_summary.py --about "two 2048-row bands of the 16384-wide plane on one GPU through j2p_tiled, exchange direct, -i 20 (tools/band_pmc.py): per launch = per band; final round-6 kernels" ${T}_fetch ${T}_write > ${T}.json
  python - <<PY
import json
d=json.load(open("${T}.json")); d["_shape"]=[16384, 2048]
json.dump(d, open("${T}.json","w"), indent=1)
print({k:v.get("hbm_bytes_per_launch") for k,v in d.items() if isinstance(v,dict) and "hbm_bytes_per_launch" in v})
PY
  rm -rf ${T}_fetch ${T}_write ${T}_stats
  # the kernel trace of the bench command once more: per-launch durations
  ( cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/r06_ld_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-host-to-host > /dev/null 2>&1 )
  F=$(find $O/r06_ld_trace -name '*kernel_trace.csv' | head -1); python tools/launch_durations.py $F | tee $O/r06_launch_durations.json | cut -c1-600; rm -rf $O/r06_ld_trace
  ;;
s)
  # randomised parity sweeps on the round's library, new seeds: as shipped, and — experiments build — with every
  # one-channel solve forced through double / half / quarter items walked bottom-up (16-row tile rows on every canvas)
  ( timeout 900 python tools/sweep_vs_ref.py 800 81 ) 2>&1 | tail -1 | tee $O/r06_final_sweeps.txt
  ( timeout 600 python tools/sweep_wide.py 300 82 ) 2>&1 | tail -1 | tee -a $O/r06_final_sweeps.txt
  ( timeout 900 python tools/sweep_tiled.py 250 83 ) 2>&1 | tail -1 | tee -a $O/r06_final_sweeps.txt
  ( timeout 600 python tools/sweep_bands.py 120 84 ) 2>&1 | tail -1 | tee -a $O/r06_final_sweeps.txt
  ( timeout 900 python tools/sweep_cli.py 60 85 ) 2>&1 | tail -1 | tee -a $O/r06_final_sweeps.txt
  export J2P_LIBRARY=$PWD/jpeg2png_amd/libjpeg2png_amd_exp.so J2P_RPW=16 J2P_ZONE_D=90 J2P_ZONE_B=70 J2P_ZONE_C=50 J2P_GRAD_REVERSE=1
  echo "experiments build, J2P_RPW=16 J2P_ZONE_D=90 J2P_ZONE_B=70 J2P_ZONE_C=50 J2P_GRAD_REVERSE=1:" | tee -a $O/r06_final_sweeps.txt
  ( timeout 900 python tools/sweep_vs_ref.py 600 86 ) 2>&1 | tail -1 | tee -a $O/r06_final_sweeps.txt
  ( timeout 600 python tools/sweep_wide.py 200 87 ) 2>&1 | tail -1 | tee -a $O/r06_final_sweeps.txt
  ( timeout 900 python tools/sweep_tiled.py 150 88 ) 2>&1 | tail -1 | tee -a $O/r06_final_sweeps.txt
  ;;
esac
