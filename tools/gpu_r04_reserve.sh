#!/bin/bash
set -o pipefail
out=gpurun_out/r04e; mkdir -p $out
for r in 0 1 2; do for m in 0 4; do
  echo "MFGP_BATCH_RESERVE=$r MFGP_MACRO=$m(0=default)"
  if [ $m = 0 ]; then MFGP_BATCH_RESERVE=$r BATCHES="3 4 6" python tools/batch_eval.py 2048 4096 8192; else MFGP_MACRO=$m MFGP_BATCH_RESERVE=$r BATCHES="3 4 6" python tools/batch_eval.py 2048 4096; fi
done; done > $out/batch_reserve.txt 2>&1; cat $out/batch_reserve.txt
