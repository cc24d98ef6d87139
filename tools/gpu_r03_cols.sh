#!/bin/bash
# A/B of the three-stream schedule (MFGP_COLS_STREAM=1), alone and with reserved CUs / the 4-wave chain kernel
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-cs}; out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_plans.py tests/test_gpu_kernels.py -x -q -m gpu > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log
[ $rc -ne 0 ] && exit $rc
for cfg in "0 0 8" "1 0 8" "1 0 4" "1 1 8" "1 1 4" "0 0 8" "1 0 8"; do
  set -- $cfg
  echo "COLS_STREAM=$1 U_RESERVE=$2 CHAIN_WAVES=$3"; MFGP_COLS_STREAM=$1 MFGP_U_RESERVE=$2 MFGP_CHAIN_WAVES=$3 python tools/time_eval.py 6144 8192 12288 16384 2>&1 | cut -c1-90
done | tee $out/time_eval_ab.txt
for cfg in "0 0 8" "1 0 8" "1 1 4"; do
  set -- $cfg
  printf "bench COLS_STREAM=$1 U_RESERVE=$2 CHAIN_WAVES=$3 "; MFGP_COLS_STREAM=$1 MFGP_U_RESERVE=$2 MFGP_CHAIN_WAVES=$3 timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['wall_ms_per_evaluation'], d['roofline']['achieved'], d['roofline']['uncontended']['avg_launch_ms'])"
done | tee $out/bench_ab.txt
cd /tmp
n=8192
MFGP_COLS_STREAM=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$n -- python3 $GRAFT_REPO_ROOT/tools/time_eval.py $n > $out/trace_$n.log 2>&1
(cd $GRAFT_REPO_ROOT && python tools/chain_account.py $out/trace_$n > $out/chain_account_$n.txt 2>&1; python tools/trace_timeline.py $out/trace_$n 0 400 > $out/timeline_$n.txt 2>&1)
cat $out/chain_account_$n.txt
rm -rf $out/trace_$n
