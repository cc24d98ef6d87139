#!/bin/bash
# One parameterised GPU session script (round 5: replaces the per-experiment gpu_r0*.sh of rounds 3-4).
#   usage (on the GPU box, through gpurun):  bash tools/gpu_session.sh <session> [outdir-name]
# Every session writes under gpurun_out/<outdir-name>/ and chains its steps with && -- a step that fails or times out ends the session.
set -o pipefail
session=${1:?session name}; out=gpurun_out/${2:-r06_$session}; mkdir -p $out
export MFGP_HW_QUEUES=${MFGP_HW_QUEUES:-2}
last_json() { python - "$1" <<'EOF'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
c = d.get("config", {})
print(sys.argv[1], "value", d["value"], "ms; n_gpus", d.get("n_gpus"), "ranks", c.get("ranks"), "rccl_ranks", c.get("rccl_ranks"),
      "frac", d.get("roofline", {}).get("frac"), "checksum", d.get("result_checksum"))
EOF
}
case $session in
deferred_k)
  # K^-1 as ONE deep-K stand-alone product per pass (MFGP_KINV_STREAM=0) against the streamed K = 128*MB chunks, in
  # the BATCHED regime (tools/batch_eval.py) and in the bench, alternating within this one call (box-to-box spread is 2-4 %)
  for rep in 1 2; do
    for ks in 1 0; do
      MFGP_KINV_STREAM=$ks BATCHES="3 4 6" timeout -k 10 300 python tools/batch_eval.py 4096 8192 > $out/batch_kinv${ks}_rep$rep.txt 2>&1 || { tail $out/batch_kinv${ks}_rep$rep.txt; exit 1; }
      echo "== MFGP_KINV_STREAM=$ks rep $rep"; grep '^N=' $out/batch_kinv${ks}_rep$rep.txt
    done
  done &&
  for rep in 1 2; do
    for ks in 1 0; do
      MFGP_KINV_STREAM=$ks timeout -k 10 300 python bench.py --no-cpu-baseline --steps 6 --warmup 1 > $out/bench_kinv${ks}_rep$rep.json 2> $out/bench_kinv${ks}_rep$rep.err || { tail $out/bench_kinv${ks}_rep$rep.err; exit 1; }
      echo "== bench MFGP_KINV_STREAM=$ks rep $rep"; last_json $out/bench_kinv${ks}_rep$rep.json
    done
  done
  ;;
rehearsal)
  # the N-rank bench at FULL size on the one-GPU box (every rank on GPU 0, RCCL over its socket transport: sharding.rehearsal_env).
  # 6 ranks is the most the box allows on its card (process guard); the 8-rank layout is rehearsed on the CPU (tests/test_bench_launcher.py)
  for n in ${RANKS:-6}; do
    # REHEARSAL_ARGS="--restarts 4" with 6 ranks: the 8-rank layout's STRUCTURE (restarts on the last ranks, ranks 1-2 idle -> a chain group
    # of 3 shares first run -> restart 0) within the 6 processes the box admits; RANKS=1 with the same arguments gives its reference line
    timeout -k 10 900 python bench.py --gpus $n $([ $n -gt 1 ] && echo --single-device) --no-cpu-baseline --steps 1 --warmup 1 ${REHEARSAL_ARGS:-} > $out/bench_n$n.json 2> $out/bench_n$n.err || { tail -30 $out/bench_n$n.err; exit 1; }
    last_json $out/bench_n$n.json
  done
  ;;
ab_define)
  # A/B of two compile-time variants of the library (AB_BASE / AB_DEFINE: -D switches through MFGP_BUILD_DEFINES; "" = the build the
  # repository ships), alternating within this one call: kernel parity first (variant build), then AB_CMD (default: one evaluation alone at
  # several sizes), batched passes and the bench under each build, twice
  build() { MFGP_BUILD_DEFINES="$1" python -m multifidelity_datafusion_gps_amd.build --force > $out/build.log 2>&1 || { tail -20 $out/build.log; exit 1; }; }
  build "${AB_DEFINE-}" &&
  timeout -k 10 800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_plans.py tests/test_gpu_parity.py -m gpu -x -q > $out/variant_tests.log 2>&1 || { tail -30 $out/variant_tests.log; build ""; exit 1; }
  tail -2 $out/variant_tests.log
  for rep in 1 2; do
    for v in base variant; do
      if [ $v = base ]; then build "${AB_BASE-}"; else build "${AB_DEFINE-}"; fi
      timeout -k 10 300 ${AB_CMD:-python tools/time_eval.py 512 1024 2048 4096 8192} > $out/cmd_${v}_rep$rep.txt 2>&1 || { tail $out/cmd_${v}_rep$rep.txt; build ""; exit 1; }
      echo "== $v rep $rep"; grep '^N=' $out/cmd_${v}_rep$rep.txt | cut -c1-200
      BATCHES="${AB_BATCHES:-4}" timeout -k 10 300 python tools/batch_eval.py ${AB_SIZES:-4096 8192} > $out/batch_${v}_rep$rep.txt 2>&1 || { tail $out/batch_${v}_rep$rep.txt; build ""; exit 1; }
      grep '^N=' $out/batch_${v}_rep$rep.txt
      if [ "${AB_BENCH:-1}" = 1 ]; then
        timeout -k 10 300 python bench.py --no-cpu-baseline --steps ${AB_STEPS:-6} --warmup 1 > $out/bench_${v}_rep$rep.json 2> $out/bench_${v}_rep$rep.err || { tail $out/bench_${v}_rep$rep.err; build ""; exit 1; }
        last_json $out/bench_${v}_rep$rep.json
      fi
      if [ -n "${AB_FIT-}" ]; then python tools/midsize_fit.py --evals 20 $AB_FIT > $out/fit_${v}_rep$rep.txt 2>&1; tail -3 $out/fit_${v}_rep$rep.txt | cut -c1-250; fi
    done
  done
  build ""
  ;;
pick)
  # selected tests, each group under its own timeout; PICK="file::test ..." (space separated pytest node ids / -k expressions are not split)
  for t in ${PICK:?}; do
    name=$(echo "$t" | tr '/:[]' '____')
    timeout -k 10 ${PICK_TIMEOUT:-600} python -m pytest "$t" -m gpu -x -q -s > $out/$name.log 2>&1; rc=$?
    tail -6 $out/$name.log
    [ $rc -eq 0 ] || exit $rc
  done
  ;;
profile)
  # the round's profiles (copied into profiles/ afterwards).  Every rocprofv3 line: the program directly after "--", PMC passes with
  # --kernel-trace only, power sampling off (bench.py samples power only with --power, and never under a profiler).
  R=$GRAFT_REPO_ROOT; out=$R/$out; export TMPDIR=/tmp; cd /tmp
  echo "== bench kernel stats"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_stats -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/bench_profiled.json 2> $out/bench_profiled.err
  f=$(find $out/bench_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/bench_kernel_stats.csv && head -14 $out/bench_kernel_stats.csv
  rm -rf $out/bench_stats
  echo "== one evaluation, kernel trace"
  for n in ${TRACE_SIZES:-4096 8192}; do
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$n -- python3 $R/tools/time_eval.py $n > $out/trace_$n.log 2>&1
    (cd $R && python3 tools/chain_account.py $out/trace_$n > $out/chain_account_$n.txt 2>&1; python3 tools/trace_timeline.py $out/trace_$n 0 400 > $out/timeline_$n.txt 2>&1; python3 tools/plan_flops.py $((n/128)) $out/timeline_$n.txt > $out/plan_flops_$n.txt 2>&1)
    cat $out/chain_account_$n.txt; tail -3 $out/plan_flops_$n.txt
    rm -rf $out/trace_$n
  done
  echo "== one batched pass (N = 8192, B = 4; N = 4096, B = 4), kernel trace"
  for nb in "8192 4" "4096 4"; do set -- $nb
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace_b_$1 -- python3 $R/tools/batch_trace.py $1 $2 3 > $out/trace_b_$1.log 2>&1
    python3 $R/tools/trace_last_pass.py $out/trace_b_$1 400 > $out/timeline_batch_$1_B$2.txt 2>&1; head -10 $out/timeline_batch_$1_B$2.txt
    rm -rf $out/trace_b_$1
  done
  echo "== PMC passes: HBM traffic"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python3 $R/tools/time_eval.py 8192 > $out/pmc_$c.log 2>&1
    f=$(find $out/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/pmc_${c}_time_eval_8192.csv
    rm -rf $out/pmc_$c
  done
  echo "== the adaptation loop's kernels (predict N* = 1 .. 64, rank-1 append at N = 8128): kernel stats, then the two traffic passes"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/adapt_stats -- python3 $R/tools/predv_once.py 8192 > $out/adapt_stats.log 2>&1
  f=$(find $out/adapt_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/adapt_kernel_stats.csv && head -16 $out/adapt_kernel_stats.csv
  rm -rf $out/adapt_stats
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/apmc_$c -- python3 $R/tools/predv_once.py 8192 > $out/apmc_$c.log 2>&1
    f=$(find $out/apmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/pmc_${c}_predv_once_8192.csv
    rm -rf $out/apmc_$c
  done
  (cd $R && python3 tools/pmc_summary.py 8192 $out/pmc.json FETCH_SIZE=$out/pmc_FETCH_SIZE_time_eval_8192.csv WRITE_SIZE=$out/pmc_WRITE_SIZE_time_eval_8192.csv ADAPT_FETCH_SIZE=$out/pmc_FETCH_SIZE_predv_once_8192.csv ADAPT_WRITE_SIZE=$out/pmc_WRITE_SIZE_predv_once_8192.csv ADAPT_STATS=$out/adapt_kernel_stats.csv > $out/pmc_summary.log 2>&1; tail -60 $out/pmc_summary.log)
  echo "== the adaptation loop's kernels: SQ counters (wait / active / matrix-pipe busy fractions)"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $out/asq -- python3 $R/tools/predv_once.py 8192 > $out/asq.log 2>&1
  (cd $R && python3 tools/sq_summary.py $out/asq > $out/adapt_sq.txt 2>&1; cat $out/adapt_sq.txt)
  rm -rf $out/asq
  echo "== PMC pass: matrix-pipe busy"
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmcA_eval_d -- python3 $R/tools/time_eval.py 8192 > $out/pmcA_eval.log 2>&1
  f=$(find $out/pmcA_eval_d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/pmcA_eval.csv
  f=$(find $out/pmcA_eval_d -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f $out/pmcA_eval_trace.csv
  rm -rf $out/pmcA_eval_d
  (cd $R && python3 tools/mfma_counters.py $out $out/mfma_counters.json > $out/mfma_counters.txt 2>&1; python3 -c "
import json; d=json.load(open('$out/mfma_counters.json'))['pmcA_eval']
for k,v in d.items(): print('%-28s launches %4d  %8.3f ms  clock %.2f GHz  mfma_busy %.3f (CU-busy basis) %.3f (launch basis)' % (k, v['launches'], v['duration_ms'], v.get('clock_ghz',0), v.get('mfma_busy',0), v.get('mfma_busy_g',0)))")
  find $out -name "*.csv" -size +6M -delete
  cd $R
  ;;
measure)
  # the un-profiled records: batched passes, mid-size fits, per-rank share of a sharded evaluation, the BASELINE configurations, small N
  python3 tools/batch_eval.py 512 1024 2048 4096 8192 > $out/batch_eval.txt 2>&1; cat $out/batch_eval.txt
  python3 tools/midsize_fit.py 64 128 256 512 1024 2048 4096 > $out/midsize_fit.txt 2>&1; cut -c1-300 $out/midsize_fit.txt
  python3 tools/midsize_fit.py --evals 20 2048 4096 >> $out/midsize_fit.txt 2>&1; tail -3 $out/midsize_fit.txt | cut -c1-300
  python3 tools/shard_projection.py 4096 8192 > $out/shard_projection.txt 2>&1; cat $out/shard_projection.txt
  timeout -k 10 500 python3 tools/run_configs.py > $out/configs.txt 2>&1; cat $out/configs.txt
  python3 tools/small_n_latency.py > $out/small_n.txt 2>&1; tail -12 $out/small_n.txt
  ;;
soak)
  # randomised parity soak against the oracle (240 s) and against the quad-precision values (300 s), and the N = 8192 truth check
  timeout -k 10 400 python3 tools/fuzz_parity.py 240 ${SOAK_SEED:-51} 3000 > $out/fuzz_parity.txt 2>&1; tail -2 $out/fuzz_parity.txt
  timeout -k 10 500 python3 tools/fuzz_parity.py 300 ${SOAK_SEED:-51} 700 truth > $out/fuzz_truth.txt 2>&1; tail -2 $out/fuzz_truth.txt
  timeout -k 10 900 python3 tools/truth_check.py 8192 > $out/truth_check.txt 2>&1; tail -12 $out/truth_check.txt
  ;;
accept)
  # what the driver runs at round end: smoke, the whole -m gpu suite, the default bench line (CPU baseline included)
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
  tail -2 $out/smoke.log
  MFGP_PARITY_ERRORS=$out/parity_errors.json timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?; tail -5 $out/tests.log; [ $rc -eq 0 ] || exit $rc
  timeout -k 10 600 python bench.py > $out/bench.json 2> $out/bench.err || { tail -30 $out/bench.err; exit 1; }
  last_json $out/bench.json
  ;;
tests)
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > $out/tests.log 2>&1; rc=$?; tail -15 $out/tests.log; exit $rc
  ;;
bench)
  timeout -k 10 600 python bench.py ${BENCH_ARGS:-} > $out/bench.json 2> $out/bench.err || { tail -30 $out/bench.err; exit 1; }
  last_json $out/bench.json
  ;;
*) echo "unknown session $session"; exit 2 ;;
esac
