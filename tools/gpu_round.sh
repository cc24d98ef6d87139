#!/bin/bash
# usage: tools/gpu_round.sh <tag> [tests|notests] -- GPU round: -m gpu suite, per-stage timings of both planners, kernel-trace
# summaries of one evaluation (N = 4096, 8192), bench.  Everything lands under gpurun_out/<tag>/.
set -o pipefail
tag=${1:-rb}; what=${2:-tests}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
if [ "$what" = tests ]; then
  echo "== pytest -m gpu"
  timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=12 > $out/gpu_tests.log 2>&1
  rc=$?; tail -15 $out/gpu_tests.log; [ $rc -ne 0 ] && echo "TESTS FAILED rc=$rc"
fi
echo "== time_eval (sweep plan)"
timeout -k 10 200 python tools/time_eval.py 128 512 1024 2048 4096 8192 16384 > $out/time_eval.txt 2>&1; cat $out/time_eval.txt
echo "== time_eval (MFGP_PLAN=levels: separate inverse + K^-1 phases)"
MFGP_PLAN=levels timeout -k 10 200 python tools/time_eval.py 1024 2048 4096 8192 16384 > $out/time_eval_levels.txt 2>&1; cat $out/time_eval_levels.txt
echo "== traces"
for n in 4096 8192; do
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace_$n -- python3 $GRAFT_REPO_ROOT/tools/time_eval.py $n > $GRAFT_REPO_ROOT/$out/trace_$n.log 2>&1)
  python tools/chain_account.py $out/trace_$n > $out/chain_account_$n.txt 2>&1
  python tools/trace_summary.py $out/trace_$n > $out/trace_summary_$n.txt 2>&1
  python tools/trace_timeline.py $out/trace_$n 0 400 > $out/timeline_$n.txt 2>&1
  cat $out/chain_account_$n.txt; head -14 $out/trace_summary_$n.txt
  find $out/trace_$n -name "*.csv" -size +20M -delete
done
echo "== bench"
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 2500 $out/bench.json; tail -3 $out/bench.err
