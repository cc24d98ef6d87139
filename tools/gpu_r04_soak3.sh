#!/bin/bash
out=gpurun_out/${1:-r04soak3}; mkdir -p $out
timeout -k 10 600 python tools/fuzz_parity.py 500 21 1200 truth > $out/fuzz_truth_seed21_n1200.txt 2>&1; rc1=$?; tail -3 $out/fuzz_truth_seed21_n1200.txt
FUZZ_NOISE_LO=1e-6 timeout -k 10 560 python tools/fuzz_parity.py 480 22 1200 truth > $out/fuzz_truth_lownoise_seed22_n1200.txt 2>&1; rc2=$?; tail -3 $out/fuzz_truth_lownoise_seed22_n1200.txt
exit $(( rc1 + rc2 ))
