#!/bin/bash
# Round 5's GPU calls, one section per call (bash tools/gpu_r05.sh <section>); everything lands under gpurun_out/r05_*.
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
case $S in
a)
  # new tests first, then the suite without the two long reference runs; then same-box A/B: levels on / off, decomposition
  ( timeout 900 python -m pytest tests/test_fineprint_gpu.py tests/test_tiled_verify_gpu.py -q -x --timeout 900 -s ) > $O/r05_a_new_tests.log 2>&1; echo "new tests rc=$?"; tail -15 $O/r05_a_new_tests.log
  ( timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -k "not full_size_i100 and not config2_4096 and not fineprint and not tiled_verify" ) > $O/r05_a_suite.log 2>&1; echo "suite rc=$?"; tail -25 $O/r05_a_suite.log
  for rep in 1 2; do
    for v in base levels0; do
      if [ $v = base ]; then $BENCH 2>/dev/null | line | brief $v; else J2P_LIBRARY=ab/libj2p_$v.so $BENCH 2>/dev/null | line | brief $v; fi
    done
  done | tee $O/r05_ab_levels.jsonl
  for v in base noarith notraffic nohalo shortdiv; do
    if [ $v = base ]; then $BENCH 2>/dev/null | line | brief $v; else J2P_LIBRARY=ab/libj2p_$v.so $BENCH 2>/dev/null | line | brief $v; fi
  done | tee $O/r05_decomposition.jsonl
  ;;
b)
  # the single-launch iteration: correctness first (fused tests + the schedule-equivalence test), then same-box A/B by size
  ( timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_tiled_verify_gpu.py -q -x --timeout 600 ) > $O/r05_b_fused_tests.log 2>&1; echo "fused tests rc=$?"; tail -12 $O/r05_b_fused_tests.log
  ( timeout 600 python -m pytest tests/test_parity_gpu.py -q -x --timeout 600 -k "schedule_switch or full_size_against or concurrent" ) > $O/r05_b_parity.log 2>&1; echo "parity subset rc=$?"; tail -5 $O/r05_b_parity.log
  timeout 600 python tools/fuse_ab.py 100 | tee $O/r05_single_launch.jsonl
  ;;
c)
  # what makes the single-launch iteration slow: timing-only variants (results wrong where noted), then the teardown fix
  for v in release fuse_plainstore fuse_plainload fuse_nowait fuse_static fuse_all; do
    if [ $v = release ]; then timeout 200 python tools/fuse_ab.py 100 1920 1080 2048 2048 4096 4096 --only old,fused; else J2P_LIBRARY=ab/libj2p_$v.so timeout 200 python tools/fuse_ab.py 100 1920 1080 2048 2048 4096 4096 --only fused; fi
  done 2>&1 | grep '^{' | tee $O/r05_single_launch_attribution.jsonl
  ( timeout 600 python -m pytest tests/test_tiled_verify_gpu.py -q -x --timeout 120 ) > $O/r05_c_verify_tests.log 2>&1; echo "verify tests rc=$?"; tail -12 $O/r05_c_verify_tests.log
  ;;
d)
  # the whole GPU suite on the pruned release library (+ experiments build for the schedule tests), smoke, then A/B: five wavefronts per SIMD
  ( timeout 1500 python -m pytest tests -m gpu -q --durations=6 --timeout 900 ) > $O/r05_d_suite.log 2>&1; echo "suite rc=$?"; tail -40 $O/r05_d_suite.log
  ( timeout 300 python __graft_entry__.py --smoke ) 2>&1 | tail -1
  for rep in 1 2; do
    for v in base waves5_ring3 ring3; do
      if [ $v = base ]; then $BENCH 2>/dev/null | line | brief $v; else J2P_LIBRARY=ab/libj2p_$v.so $BENCH 2>/dev/null | line | brief $v; fi
    done
  done | tee $O/r05_ab_waves5.jsonl
  ;;
e)
  # the two tests the prune broke, then five wavefronts per SIMD for the hot gradient kernel by size (same box, alternating)
  ( timeout 900 python -m pytest tests/test_tiled_verify_gpu.py tests/test_parity_gpu.py -q --timeout 600 -k "picker or broken or randomised_band_splits" ) > $O/r05_e_tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/r05_e_tests.log
  for sz in "1920 1080" "2048 2048" "4096 4096" "16384 2048"; do
    set -- $sz
    for v in base hot5 base hot5; do
      L=""; [ $v = hot5 ] && L=ab/libj2p_hot5.so
      ( J2P_LIBRARY=$L timeout 200 python bench.py --size $1 --height $2 --iterations 100 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-to-host ) 2>/dev/null | line | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'plane': '$1x$2', 'variant': '$v', 'us_per_iteration': round(r['iteration_ms']*1e3,2), 'k_gradient_us': round(r['per_kernel']['k_gradient']['avg_launch_ms']*1e3,1)}))"
    done
  done | tee $O/r05_ab_hot5_by_size.jsonl
  ;;
f)
  # why host-to-host went from 66 to 98 ms: the split per call — as is, without the huge-page output planes, without the arena pool;
  # then the kernels of that path under rocprofv3
  timeout 200 python tools/h2h_probe.py 5 2>/dev/null | tee $O/r05_h2h_probe.jsonl
  J2P_LIBRARY=ab/libj2p_nohuge.so timeout 200 python tools/h2h_probe.py 5 2>/dev/null | tee -a $O/r05_h2h_probe.jsonl
  J2P_POOL_MIB=0 timeout 200 python tools/h2h_probe.py 4 2>/dev/null | tee -a $O/r05_h2h_probe.jsonl
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/r05_h2h_stats -- python $GRAFT_REPO_ROOT/tools/h2h_probe.py 3 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
  find $O/r05_h2h_stats -name '*kernel_stats.csv' -exec head -5 {} \;
  find $O/r05_h2h_stats -name '*.csv' -size +1M -delete
  ;;
g)
  # host-to-host after the fix, then the default bench line (what the driver runs) for the record
  timeout 200 python tools/h2h_probe.py 6 2>/dev/null | tee $O/r05_h2h_probe_fixed.jsonl
  ( timeout 600 python bench.py --gpus 1 --steps 25 --warmup 3 ) 2>/dev/null | line > $O/r05_bench_driver_shape.json; cut -c1-600 $O/r05_bench_driver_shape.json
  ( timeout 600 python bench.py ) 2>/dev/null | line > $O/r05_bench_n1.json; python -c "
import json; d=json.load(open('$O/r05_bench_n1.json')); print(d['value'], d['host_to_host']['ms_per_call'], d['host_to_host']['split_ms'])"
  ;;
h)
  # which context makes the stall come back, and which piece of the helper thread's work it follows
  for ctx in "" "--torch" "--resident" "--torch --resident"; do timeout 200 python tools/h2h_probe.py 4 $ctx 2>/dev/null; done | tee $O/r05_h2h_context.jsonl
  for v in SKIP_TOUCH SKIP_FREE; do J2P_LIBRARY=ab/libj2p_hk_$v.so timeout 200 python tools/h2h_probe.py 4 --torch --resident 2>/dev/null; done | tee -a $O/r05_h2h_context.jsonl
  ;;
i)
  for ctx in "--resident-nodl" "--resident-small" "--resident"; do timeout 200 python tools/h2h_probe.py 4 $ctx 2>/dev/null; done | tee $O/r05_h2h_context2.jsonl
  J2P_POOL_MIB=0 timeout 200 python tools/h2h_probe.py 4 --resident 2>/dev/null | sed 's/"--resident"/"--resident, J2P_POOL_MIB=0"/' | tee -a $O/r05_h2h_context2.jsonl
  timeout 200 python tools/h2h_probe.py 10 2>/dev/null | sed 's/"plain"/"plain, 10 calls"/' | tee -a $O/r05_h2h_context2.jsonl
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/r05_h2h_trace2 -- python $GRAFT_REPO_ROOT/tools/h2h_probe.py 3 --resident > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
  python - <<PY
import csv, glob
f = glob.glob("$O/r05_h2h_trace2/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:30]) for r in rows)
inits = [i for i, k in enumerate(ks) if "k_init_state" in k[2]]
for ci, i0 in enumerate(inits):
    i1 = inits[ci + 1] if ci + 1 < len(inits) else len(ks)
    g = [k for k in ks[i0:i1] if "k_gradient" in k[2]]
    print("solve", ci, "gradients", len(g), "long:", [(j, round((k[0] - g[0][0]) / 1e6, 2), round((k[1] - k[0]) / 1e6, 2)) for j, k in enumerate(g) if k[1] - k[0] > 1e6])
PY
  find $O/r05_h2h_trace2 -name '*.csv' -delete
  ;;
j)
  for v in BOTH NO_THREAD AFTER_SYNC; do J2P_LIBRARY=ab/libj2p_hk_$v.so timeout 200 python tools/h2h_probe.py 4 --resident-nodl 2>/dev/null; done | tee $O/r05_h2h_context3.jsonl
  ;;
k)
  for ctx in "" "--resident-nodl" "--torch --resident"; do timeout 200 python tools/h2h_probe.py 5 $ctx 2>/dev/null; done | tee $O/r05_h2h_final.jsonl
  ( timeout 300 python -m pytest tests/test_fineprint_gpu.py tests/test_parity_gpu.py -q --timeout 300 -k "fineprint or drop_in or concurrent" ) 2>&1 | tail -3
  ( timeout 600 python bench.py ) 2>/dev/null | line > $O/r05_bench_n1.json; python -c "
import json; d=json.load(open('$O/r05_bench_n1.json')); print(d['value'], d['host_to_host']['ms_per_call'], d['host_to_host']['ms_per_call_all'], d['host_to_host']['split_ms'])"
  ;;
l)
  for ctx in "" "--torch --resident"; do timeout 200 python tools/h2h_probe.py 5 $ctx 2>/dev/null; done | tee $O/r05_h2h_final.jsonl
  ( timeout 900 python -m pytest tests/test_fineprint_gpu.py tests/test_parity_gpu.py tests/test_cli.py tests/test_capi.py -q --timeout 600 -k "fineprint or drop_in or concurrent or cli or reference_program or capi" -m gpu ) 2>&1 | tail -4
  ( timeout 600 python bench.py ) 2>/dev/null | line > $O/r05_bench_n1.json; python -c "
import json; d=json.load(open('$O/r05_bench_n1.json')); print(d['value'], d['host_to_host']['ms_per_call'], d['host_to_host']['ms_per_call_all'], d['host_to_host']['split_ms'])"
  ( timeout 600 python bench.py --gpus 1 --steps 25 --warmup 3 ) 2>/dev/null | line > $O/r05_bench_driver_shape.json; python -c "
import json; d=json.load(open('$O/r05_bench_driver_shape.json')); print('driver shape', d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity']['bit_identical'], d['host_to_host']['ms_per_call'])"
  ;;
m)
  # the round's last word: the whole suite, smoke, and the bench as the driver runs it, on the final tree
  ( timeout 1500 python -m pytest tests -m gpu -q --durations=6 --timeout 900 ) > $O/r05_m_suite.log 2>&1; echo "suite rc=$?"; tail -50 $O/r05_m_suite.log
  ( timeout 300 python __graft_entry__.py --smoke ) 2>&1 | tail -1
  ( timeout 600 python bench.py --gpus 1 --steps 25 --warmup 3 ) 2>/dev/null | line > $O/r05_bench_driver_shape.json; python -c "
import json; d=json.load(open('$O/r05_bench_driver_shape.json')); print('driver shape', d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity']['bit_identical'], d['host_to_host']['ms_per_call'], d['host_to_host']['split_ms'])"
  ( timeout 600 python bench.py ) 2>/dev/null | line > $O/r05_bench_n1.json; python -c "
import json; d=json.load(open('$O/r05_bench_n1.json')); print('default', d['value'], d['host_to_host']['ms_per_call'])"
  ;;
n)
  ( timeout 600 python -m pytest tests/test_fineprint_gpu.py tests/test_capi.py -q --timeout 300 -m gpu ) 2>&1 | tail -3
  ( timeout 600 python bench.py --gpus 1 --steps 25 --warmup 3 ) 2>/dev/null | line > $O/r05_bench_driver_shape.json; python -c "
import json; d=json.load(open('$O/r05_bench_driver_shape.json')); r=d['roofline']; print('driver shape', d['value'], d['ms_per_step'], r['frac'], r['event_pair_overhead_us'], {k: (v['avg_launch_ms'], v['frac']) for k, v in r['per_kernel'].items()}, d['parity']['bit_identical'], d['host_to_host']['ms_per_call'])"
  ;;
o)
  ( timeout 1200 python -m pytest tests/test_tiled_verify_gpu.py tests/test_tiled_c_gpu.py tests/test_batch_gpu.py -q --timeout 600 -m gpu ) > $O/r05_o_tiled_tests.log 2>&1; echo "tiled tests rc=$?"; grep -E "passed|failed|Error" $O/r05_o_tiled_tests.log | tail -8
  ;;
p)
  timeout 300 python tools/batch_prealloc.py 128 8 2>/dev/null | tee $O/r05_batch_prealloc.jsonl
  ;;
q)
  # planes beyond the Infinity Cache (NT >= 1): five wavefronts + ring of three (the library now) against four + four (big4), same box
  for sz in "4096 4096 500" "8192 4096 100" "16384 2048 100" "8192 8192 100"; do
    set -- $sz
    for v in new big4 new big4; do
      L=""; [ $v = big4 ] && L=ab/libj2p_big4.so
      ( J2P_LIBRARY=$L timeout 300 python bench.py --size $1 --height $2 --iterations $3 --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --no-host-to-host ) 2>/dev/null | line | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'plane': '$1x$2 -i $3', 'variant': '$v', 'us_per_iteration': round(r['iteration_ms']*1e3,2), 'k_gradient_us': round(r['per_kernel']['k_gradient']['avg_launch_ms']*1e3,1), 'parity': (d.get('parity') or {}).get('bit_identical')}))"
    done
  done | tee $O/r05_ab_big_planes.jsonl
  ;;
r)
  ( timeout 300 python bench.py --config batch --steps 2 --warmup 1 --batch 64 ) 2>&1 | line | tee $O/r05_bench_batch.json | cut -c1-300
  ( timeout 600 python bench.py --force-tiled --bands 8 --steps 2 --warmup 1 --no-cpu-baseline ) 2>&1 | line > $O/r05_bench_tiled_8bands_1gpu.json; python -c "
import json; d=json.load(open('$O/r05_bench_tiled_8bands_1gpu.json')); print(d['value'], d['config']['engine'], d['parity']['bit_identical']); [print(o.get('config','')[:50], o.get('Mpx_it_per_s'), o.get('images_per_s')) for o in d['other_configs']]"
  ;;
s)
  # randomised parity sweeps on the round's final library, new seeds
  ( timeout 900 python tools/sweep_vs_ref.py 800 71 ) 2>&1 | tail -1 | tee $O/r05_final_sweeps.txt
  ( timeout 600 python tools/sweep_wide.py 300 72 ) 2>&1 | tail -1 | tee -a $O/r05_final_sweeps.txt
  ( timeout 900 python tools/sweep_tiled.py 250 73 ) 2>&1 | tail -1 | tee -a $O/r05_final_sweeps.txt
  ( timeout 600 python tools/sweep_bands.py 120 74 ) 2>&1 | tail -1 | tee -a $O/r05_final_sweeps.txt
  ( timeout 900 python tools/sweep_cli.py 60 75 ) 2>&1 | tail -1 | tee -a $O/r05_final_sweeps.txt
  ;;
t)
  # batch engine: a stream per image (old) against the worker's own streams (new), outputs in a ring; same box, alternating
  for v in new old new old; do
    L=""; [ $v = old ] && L=ab/libj2p_batch_old.so
    J2P_LIBRARY=$L timeout 300 python tools/batch_prealloc.py 192 8 2>/dev/null | grep ring | sed "s/\"outputs\"/\"streams\": \"$v\", \"outputs\"/"
  done | tee $O/r05_batch_streams.jsonl
  ( timeout 600 python -m pytest tests/test_batch_gpu.py tests/test_baseline_configs_gpu.py -q --timeout 600 -m gpu -k "batch" ) 2>&1 | grep -E "passed|failed"
  ;;
esac
