#!/bin/bash
out=gpurun_out/${1:-r04soak2}; mkdir -p $out
FUZZ_NOISE_LO=1e-6 timeout -k 10 420 python tools/fuzz_parity.py 300 12 700 truth > $out/fuzz_truth_lownoise_seed12.txt 2>&1; rc=$?; tail -3 $out/fuzz_truth_lownoise_seed12.txt
FUZZ_NOISE_LO=1e-6 timeout -k 10 300 python tools/fuzz_parity.py 200 13 3000 > $out/fuzz_parity_lownoise_seed13.txt 2>&1; rc2=$?; tail -1 $out/fuzz_parity_lownoise_seed13.txt
exit $(( rc + rc2 ))
