#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-ql}; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -x -q > $out/tests.log 2>&1; tail -3 $out/tests.log
MFGP_LEAF_STAMPS=1 timeout -k 10 60 python tools/leaf_stamps.py 2>&1 | tail -2
timeout -k 10 120 python tools/time_eval.py 128 512 1024 2048 4096 8192 2>&1 | awk '{printf "  %s %s", $1, $3}'; echo
MFGP_LEAF=2 timeout -k 10 120 python tools/time_eval.py 128 512 1024 2048 4096 2>&1 | awk '{printf "  %s %s", $1, $3}'; echo
