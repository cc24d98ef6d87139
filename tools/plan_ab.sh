#!/bin/bash
# A/B of planner switches: one objective+gradient evaluation alone on the GPU (tools/time_eval.py), total ms per size
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-ab}; mkdir -p $out
run() { # label, env, sizes
  printf "%-44s" "$1"
  env $2 timeout -k 10 120 python tools/time_eval.py $3 2>&1 | awk '{printf "  %s %s", $1, $3}' ; echo
}
{
run "default" "A=1" "1024 2048 3072 4096 6144 8192 16384"
run "PLAN=levels" "MFGP_PLAN=levels" "1024 2048 3072 4096 6144 8192 16384"
run "MACRO=2" "MFGP_MACRO=2" "2048 4096"
run "MACRO=6" "MFGP_MACRO=6" "4096 6144 8192"
run "MACRO=8" "MFGP_MACRO=8" "6144 8192 16384"
run "BULK_EVERY=1" "MFGP_BULK_EVERY=1" "8192"
run "BULK_EVERY=4" "MFGP_BULK_EVERY=4" "8192 16384"
run "MACRO=4 BULK_EVERY=4" "MFGP_MACRO=4 MFGP_BULK_EVERY=4" "8192"
run "MACRO=4 BULK_EVERY=1" "MFGP_MACRO=4 MFGP_BULK_EVERY=1" "6144 8192"
run "KINV_STREAM=0" "MFGP_KINV_STREAM=0" "4096 8192"
run "MACRO=8 KINV_STREAM=0" "MFGP_MACRO=8 MFGP_KINV_STREAM=0" "8192 16384"
run "XPANEL_MERGE=0" "MFGP_XPANEL_MERGE=0" "8192"
run "XPANEL_MERGE=1" "MFGP_XPANEL_MERGE=1" "4096"
run "BULK_EVERY=2" "MFGP_BULK_EVERY=2" "4096"
run "MACRO=16" "MFGP_MACRO=16" "16384"
} | tee $out/plan_ab.txt
