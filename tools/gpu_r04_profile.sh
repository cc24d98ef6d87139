#!/bin/bash
# Round-4 profiles (copied into profiles/ afterwards).  Every rocprofv3 line: the program directly after "--", PMC passes with
# --kernel-trace only, power sampling off (bench.py samples power only with --power, and never under a profiler).
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r04p}; out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
echo "== bench kernel stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_stats -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/bench_profiled.json 2> $out/bench_profiled.err
f=$(find $out/bench_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/bench_kernel_stats.csv && head -14 $out/bench_kernel_stats.csv
rm -rf $out/bench_stats
echo "== one evaluation, kernel trace"
for n in 4096 8192; do
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
(cd $R && python3 tools/pmc_summary.py 8192 $out/pmc.json FETCH_SIZE=$out/pmc_FETCH_SIZE_time_eval_8192.csv WRITE_SIZE=$out/pmc_WRITE_SIZE_time_eval_8192.csv > $out/pmc_summary.log 2>&1; tail -40 $out/pmc_summary.log)
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
echo "== batched passes, lanes, mid-size fits, per-rank share of a sharded evaluation"
python3 tools/batch_eval.py 512 1024 2048 4096 8192 > $out/batch_eval.txt 2>&1; cat $out/batch_eval.txt
for n in 1024 2048 4096 8192; do python3 tools/lanes_batch.py $n "1:1 1:2 1:3 1:4 1:6 2:2 2:3 3:2"; done > $out/lanes_batch.txt 2>&1; cat $out/lanes_batch.txt
python3 tools/midsize_fit.py 256 512 1024 2048 4096 > $out/midsize_fit.txt 2>&1; cat $out/midsize_fit.txt
python3 tools/midsize_fit.py --evals 20 2048 4096 >> $out/midsize_fit.txt 2>&1; tail -4 $out/midsize_fit.txt
python3 tools/shard_projection.py 4096 8192 16384 > $out/shard_projection.txt 2>&1; cat $out/shard_projection.txt
echo "== configs"
timeout -k 10 500 python3 tools/run_configs.py > $out/configs.txt 2>&1; cat $out/configs.txt
echo "== randomised parity soak (240 s)"
timeout -k 10 400 python3 tools/fuzz_parity.py 240 0 3000 > $out/fuzz_parity.txt 2>&1; tail -2 $out/fuzz_parity.txt
