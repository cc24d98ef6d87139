#!/bin/bash
# round 4: batched evaluation + lock-stepped restarts -- parity first, then the timings
set -o pipefail
mkdir -p gpurun_out/r04b
python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py tests/test_gpu_plans.py -m gpu -x -q > gpurun_out/r04b/tests.log 2>&1 || { tail -30 gpurun_out/r04b/tests.log; exit 1; }
tail -3 gpurun_out/r04b/tests.log
python tools/batch_eval.py 512 1024 2048 4096 8192 > gpurun_out/r04b/batch_eval.txt 2>&1 && cat gpurun_out/r04b/batch_eval.txt
python tools/midsize_fit.py 256 1024 2048 4096 > gpurun_out/r04b/midsize_fit.txt 2>&1 && cat gpurun_out/r04b/midsize_fit.txt
