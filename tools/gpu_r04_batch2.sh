#!/bin/bash
set -o pipefail
out=gpurun_out/r04d; mkdir -p $out
python -m pytest tests/test_gpu_models.py tests/test_gpu_plans.py -m gpu -x -q -k "lockstep or batched or planner_variant" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
BATCHES="1 3 4 6" python tools/batch_eval.py 1024 2048 4096 8192 > $out/batch_eval_default.txt 2>&1; cat $out/batch_eval_default.txt
for m in 2 3 4 6; do echo "MFGP_MACRO=$m"; MFGP_MACRO=$m BATCHES="1 4 6" python tools/batch_eval.py 2048 4096; done > $out/batch_eval_macro_mid.txt 2>&1; cat $out/batch_eval_macro_mid.txt
for m in 4 6 8; do echo "MFGP_MACRO=$m"; MFGP_MACRO=$m BATCHES="1 3 6" python tools/batch_eval.py 8192; done > $out/batch_eval_macro_8192.txt 2>&1; cat $out/batch_eval_macro_8192.txt
python tools/midsize_fit.py 1024 2048 4096 > $out/midsize_fit.txt 2>&1; cat $out/midsize_fit.txt
