#!/bin/bash
# quick GPU round: kernel / parity / plan tests, one-evaluation timings, kernel traces with chain accounting.
# usage: gpu_r03_quick.sh <tag> [pytest|nopytest] [sizes...]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-q}; out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
mode=${2:-pytest}
export TMPDIR=/tmp
if [ "$mode" = "pytest" ]; then
  timeout -k 10 700 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_plans.py -x -q -m gpu > $out/pytest.log 2>&1
  rc=$?; tail -4 $out/pytest.log
  [ $rc -ne 0 ] && exit $rc
fi
python tools/time_eval.py 1024 2048 4096 6144 8192 16384 2>&1 | tee $out/time_eval.txt
cd /tmp
for n in 4096 8192; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$n -- python3 $GRAFT_REPO_ROOT/tools/time_eval.py $n > $out/trace_$n.log 2>&1
  (cd $GRAFT_REPO_ROOT && python tools/chain_account.py $out/trace_$n > $out/chain_account_$n.txt 2>&1; python tools/trace_summary.py $out/trace_$n > $out/trace_summary_$n.txt 2>&1; python tools/trace_timeline.py $out/trace_$n 0 400 > $out/timeline_$n.txt 2>&1; python tools/plan_flops.py $((n/128)) $out/timeline_$n.txt > $out/plan_flops_$n.txt 2>&1)
  cat $out/chain_account_$n.txt; head -9 $out/trace_summary_$n.txt; tail -3 $out/plan_flops_$n.txt
  rm -rf $out/trace_$n
done
