#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab}; mkdir -p $out
run() { printf "%-44s" "$1"; env $2 timeout -k 10 120 python tools/time_eval.py $3 2>&1 | awk '{printf "  %s %s", $1, $3}' ; echo; }
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_plans.py -m gpu -x -q > $out/tests.log 2>&1; tail -4 $out/tests.log
{
run "chain32 register-direct (default)" "A=1" "256 512 1024 2048 3072 4096"
run "chain32 LDS (CHAIN32_DIRECT=0)" "MFGP_CHAIN32_DIRECT=0" "256 512 1024 2048 3072 4096"
run "chain32 register-direct (repeat)" "A=1" "256 512 1024 2048 3072 4096"
} | tee $out/plan_ab2.txt
