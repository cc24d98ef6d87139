#!/bin/bash
# GPU round A: full -m gpu suite, per-stage timings, kernel-trace timelines of one evaluation (N = 4096, 8192), bench.
set -o pipefail
mkdir -p gpurun_out/ra
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
echo "== pytest -m gpu" 
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/ra/gpu_tests.log 2>&1
rc=$?
tail -25 gpurun_out/ra/gpu_tests.log
[ $rc -ne 0 ] && echo "TESTS FAILED rc=$rc"
echo "== time_eval"
timeout -k 10 200 python tools/time_eval.py 512 1024 2048 4096 8192 16384 > gpurun_out/ra/time_eval.txt 2>&1 && cat gpurun_out/ra/time_eval.txt
echo "== traces"
for n in 4096 8192; do
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ra/trace_$n -- python3 $GRAFT_REPO_ROOT/tools/time_eval.py $n > $GRAFT_REPO_ROOT/gpurun_out/ra/trace_$n.log 2>&1)
  python tools/chain_account.py gpurun_out/ra/trace_$n > gpurun_out/ra/chain_account_$n.txt 2>&1
  python tools/trace_summary.py gpurun_out/ra/trace_$n > gpurun_out/ra/trace_summary_$n.txt 2>&1
  python tools/trace_timeline.py gpurun_out/ra/trace_$n 0 400 > gpurun_out/ra/timeline_$n.txt 2>&1
  cat gpurun_out/ra/chain_account_$n.txt
  # the raw traces are large: keep the summaries only
  find gpurun_out/ra/trace_$n -name "*.csv" -size +20M -delete
done
echo "== bench"
timeout -k 10 400 python bench.py > gpurun_out/ra/bench.json 2> gpurun_out/ra/bench.err; tail -c 3000 gpurun_out/ra/bench.json
