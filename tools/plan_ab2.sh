#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab}; mkdir -p $out
run() { printf "%-44s" "$1"; env $2 timeout -k 10 200 python tools/time_eval.py $3 2>&1 | awk '{printf "  %s %s", $1, $3}' ; echo; }
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; tail -4 $out/tests.log
{
S="128 256 384 512 640 768 896 1024"
run "default" "A=1" "$S"
run "KINV_ON_CHAIN=0" "MFGP_KINV_ON_CHAIN=0" "$S"
} | tee $out/plan_ab2.txt
python tools/small_n_latency.py 2>&1 | tail -8
