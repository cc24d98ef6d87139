#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab}; mkdir -p $out
run() { printf "%-44s" "$1"; env $2 timeout -k 10 120 python tools/time_eval.py $3 2>&1 | awk '{printf "  %s %s", $1, $3}' ; echo; }
{
run "chain32 K16 single (default)" "A=1" "512 1024 2048 4096"
run "chain32 K32 single" "MFGP_CHAIN32=1" "512 1024 2048 4096"
run "chain32 K16 double" "MFGP_CHAIN32=2" "512 1024 2048 4096"
run "chain32 K16 single (repeat)" "A=1" "512 1024 2048 4096"
} | tee $out/plan_ab2.txt
