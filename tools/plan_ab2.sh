#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab}; mkdir -p $out
run() { printf "%-44s" "$1"; env $2 timeout -k 10 200 python tools/time_eval.py $3 2>&1 | awk '{printf "  %s %s", $1, $3}' ; echo; }
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; tail -3 $out/tests.log
{
S="1024 1152 1280 1536 1664 1792 2048"
run "default" "A=1" "$S"
} | tee $out/plan_ab2.txt
