#!/bin/bash
# Round-3 profiles (copied into profiles/ afterwards): rocprofv3 kernel stats of the bench workload; ONE evaluation at
# N = 4096 / 8192 under a kernel trace (chain accounting, per-launch planned flops); PMC passes over tools/time_eval.py 8192:
# FETCH_SIZE, WRITE_SIZE (HBM traffic) and the SQ MFMA-busy set, each in its own run with --kernel-trace only and the
# program directly after "--"; the un-profiled bench line with the CPU baseline; the five BASELINE configurations.
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r03p}; out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
echo "== bench kernel stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_stats -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/bench_profiled.json 2> $out/bench_profiled.err
f=$(find $out/bench_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/bench_kernel_stats.csv && head -12 $out/bench_kernel_stats.csv
rm -rf $out/bench_stats
echo "== one evaluation, kernel trace"
for n in 4096 8192; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$n -- python3 $R/tools/time_eval.py $n > $out/trace_$n.log 2>&1
  (cd $R && python3 tools/chain_account.py $out/trace_$n > $out/chain_account_$n.txt 2>&1; python3 tools/trace_summary.py $out/trace_$n > $out/trace_summary_$n.txt 2>&1; python3 tools/trace_timeline.py $out/trace_$n 0 400 > $out/timeline_$n.txt 2>&1; python3 tools/plan_flops.py $((n/128)) $out/timeline_$n.txt > $out/plan_flops_$n.txt 2>&1)
  cat $out/chain_account_$n.txt; head -9 $out/trace_summary_$n.txt; tail -3 $out/plan_flops_$n.txt
  rm -rf $out/trace_$n
done
echo "== PMC passes: HBM traffic"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python3 $R/tools/time_eval.py 8192 > $out/pmc_$c.log 2>&1
  f=$(find $out/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/pmc_${c}_time_eval_8192.csv
  rm -rf $out/pmc_$c
done
(cd $R && python3 tools/pmc_summary.py 8192 $out/pmc.json FETCH_SIZE=$out/pmc_FETCH_SIZE_time_eval_8192.csv WRITE_SIZE=$out/pmc_WRITE_SIZE_time_eval_8192.csv > $out/pmc_summary.log 2>&1; tail -30 $out/pmc_summary.log)
echo "== PMC pass: matrix-pipe busy"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmcA_eval_d -- python3 $R/tools/time_eval.py 8192 > $out/pmcA_eval.log 2>&1
f=$(find $out/pmcA_eval_d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/pmcA_eval.csv
f=$(find $out/pmcA_eval_d -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f $out/pmcA_eval_trace.csv
rm -rf $out/pmcA_eval_d
(cd $R && python3 tools/mfma_counters.py $out $out/mfma_counters.json > $out/mfma_counters.txt 2>&1; python3 -c "
import json; d=json.load(open('$out/mfma_counters.json'))['pmcA_eval']
for k,v in d.items(): print('%-28s launches %4d  %8.3f ms  clock %.2f GHz  mfma_busy %.3f (CU-busy basis) %.3f (launch basis)' % (k, v['launches'], v['duration_ms'], v.get('clock_ghz',0), v.get('mfma_busy',0), v.get('mfma_busy_g',0)))")
find $out -name "*.csv" -size +6M -delete
echo "== bench (un-profiled, with CPU baseline)"
cd $R && timeout -k 10 600 python3 bench.py > $out/bench.json 2> $out/bench.err; tail -c 2500 $out/bench.json
echo "== configs"
timeout -k 10 500 python3 tools/run_configs.py > $out/configs.txt 2>&1; cat $out/configs.txt
