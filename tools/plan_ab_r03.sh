#!/bin/bash
# round 3 A/B of the remaining planner switches under the new defaults: one evaluation alone, total ms
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r03ab}; mkdir -p $out
run() { printf "%-40s" "$1"; env $2 timeout -k 10 200 python tools/time_eval.py $3 2>&1 | awk '{printf "  %s %s", $1, $3}'; echo; }
S="2048 4096 6144 8192 16384"
{
run "default" "A=1" "$S"
run "SHIFT=1" "MFGP_SHIFT=1" "$S"
run "SHIFT=0" "MFGP_SHIFT=0" "$S"
run "BULK_EVERY=2" "MFGP_BULK_EVERY=2" "$S"
run "BULK_EVERY=3" "MFGP_BULK_EVERY=3" "6144 8192 16384"
run "XPANEL_MERGE=0" "MFGP_XPANEL_MERGE=0" "$S"
run "XPANEL_MERGE=1" "MFGP_XPANEL_MERGE=1" "$S"
run "CHAIN_SLIM=0" "MFGP_CHAIN_SLIM=0" "$S"
run "CHAIN_TILE=32" "MFGP_CHAIN_TILE=32" "6144 8192 16384"
run "CHAIN_TILE=64" "MFGP_CHAIN_TILE=64" "2048 4096"
run "T128_MIN=100" "MFGP_T128_MIN=100" "$S"
run "T128_MIN=2000" "MFGP_T128_MIN=2000" "$S"
run "KINV_STREAM=0" "MFGP_KINV_STREAM=0" "4096 8192"
run "default (repeat)" "A=1" "$S"
} | tee $out/plan_ab.txt
