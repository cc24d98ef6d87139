#!/bin/bash
# Round profiles: rocprofv3 kernel stats of the bench workload, kernel-trace summaries / timelines of ONE evaluation at
# N = 4096 and 8192, and the two PMC passes (HBM traffic).  Output under gpurun_out/<tag>/ (copy what is to be judged into profiles/).
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-prof}; out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
echo "== bench kernel stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/bench_profiled.json 2> $out/bench_profiled.err
f=$(find $out/bench_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/bench_kernel_stats.csv && head -14 $out/bench_kernel_stats.csv
find $out/bench_stats -name "*.csv" -size +5M -delete
echo "== one evaluation, kernel trace"
for n in 4096 8192; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$n -- python3 $GRAFT_REPO_ROOT/tools/time_eval.py $n > $out/trace_$n.log 2>&1
  (cd $GRAFT_REPO_ROOT && python tools/chain_account.py $out/trace_$n > $out/chain_account_$n.txt 2>&1; python tools/trace_summary.py $out/trace_$n > $out/trace_summary_$n.txt 2>&1; python tools/trace_timeline.py $out/trace_$n 0 400 > $out/timeline_$n.txt 2>&1; python tools/plan_flops.py $((n/128)) $out/timeline_$n.txt > $out/plan_flops_$n.txt 2>&1)
  cat $out/chain_account_$n.txt; head -9 $out/trace_summary_$n.txt; tail -2 $out/plan_flops_$n.txt
  find $out/trace_$n -name "*.csv" -size +20M -delete
done
echo "== PMC passes"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python3 $GRAFT_REPO_ROOT/tools/time_eval.py 8192 > $out/pmc_$c.log 2>&1
  f=$(find $out/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/pmc_${c}_time_eval_8192.csv
done
cd $GRAFT_REPO_ROOT && python tools/pmc_summary.py 8192 $out/pmc.json FETCH_SIZE=$out/pmc_FETCH_SIZE_time_eval_8192.csv WRITE_SIZE=$out/pmc_WRITE_SIZE_time_eval_8192.csv > $out/pmc_summary.log 2>&1; tail -40 $out/pmc_summary.log
find $out -name "*counter_collection.csv" -size +20M -delete
echo "== bench (unprofiled, with CPU baseline)"
timeout -k 10 500 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 1500 $out/bench.json
echo "== configs"
timeout -k 10 500 python tools/run_configs.py > $out/configs.txt 2>&1; cat $out/configs.txt
