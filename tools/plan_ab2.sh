#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab}; mkdir -p $out
run() { printf "%-44s" "$1"; env $2 timeout -k 10 200 python tools/time_eval.py $3 2>&1 | awk '{printf "  %s %s", $1, $3}' ; echo; }
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plans.py tests/test_gpu_kernels.py -m gpu -x -q > $out/tests.log 2>&1; tail -4 $out/tests.log
{
S="1024 1536 2048 3072 3584 4096 4608 5120 5632 6144"
run "default" "A=1" "$S"
run "SHIFT=0" "MFGP_SHIFT=0" "$S"
run "CHAIN_SLIM=1" "MFGP_CHAIN_SLIM=1" "$S"
run "CHAIN_SLIM=0" "MFGP_CHAIN_SLIM=0" "$S"
run "CHAIN_TILE=64" "MFGP_CHAIN_TILE=64" "$S"
run "T128_MIN=200" "MFGP_T128_MIN=200" "$S"
run "T128_MIN=450" "MFGP_T128_MIN=450" "$S"
run "BULK_EVERY=2" "MFGP_BULK_EVERY=2" "$S"
run "default" "A=1" "$S"
} | tee $out/plan_ab2.txt
