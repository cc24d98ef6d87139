#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-tr}; out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for n in 2048 4096 8192; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$n -- python3 $GRAFT_REPO_ROOT/tools/time_eval.py $n > $out/trace_$n.log 2>&1
  (cd $GRAFT_REPO_ROOT && python tools/chain_account.py $out/trace_$n > $out/chain_account_$n.txt 2>&1; python tools/trace_summary.py $out/trace_$n > $out/trace_summary_$n.txt 2>&1; python tools/trace_timeline.py $out/trace_$n 0 400 > $out/timeline_$n.txt 2>&1; python tools/plan_flops.py $((n/128)) $out/timeline_$n.txt > $out/plan_flops_$n.txt 2>&1)
  cat $out/chain_account_$n.txt; head -9 $out/trace_summary_$n.txt; tail -2 $out/plan_flops_$n.txt
  find $out/trace_$n -name "*.csv" -size +20M -delete
done
