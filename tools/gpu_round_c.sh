#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/rc; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -m gpu -x -q > $out/tests.log 2>&1; tail -3 $out/tests.log
bash tools/plan_ab.sh rc
for n in 4096 8192; do
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace_$n -- python3 $GRAFT_REPO_ROOT/tools/time_eval.py $n > $GRAFT_REPO_ROOT/$out/trace_$n.log 2>&1)
  python tools/chain_account.py $out/trace_$n > $out/chain_account_$n.txt 2>&1
  python tools/trace_summary.py $out/trace_$n > $out/trace_summary_$n.txt 2>&1
  python tools/trace_timeline.py $out/trace_$n 0 400 > $out/timeline_$n.txt 2>&1
  cat $out/chain_account_$n.txt; head -8 $out/trace_summary_$n.txt
  find $out/trace_$n -name "*.csv" -size +20M -delete
done
for c in 1 2; do
echo "== bench N=1 concurrency $c"
timeout -k 10 400 python bench.py --concurrency $c --no-cpu-baseline > $out/bench_c$c.json 2> $out/bench.err; python -c "
import json,sys; d=json.loads(open('$out/bench_c$c.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['wall_ms_per_evaluation'], d['roofline']['achieved'], d['roofline']['uncontended'])"
done
