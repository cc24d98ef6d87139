#!/bin/bash
# round 4, second session: longer soaks (GPU minutes that would otherwise lapse): against the quad-precision values, and against the oracle
# at larger sizes / in the low-noise regime
out=gpurun_out/${1:-r04soak}; mkdir -p $out
timeout -k 10 420 python tools/fuzz_parity.py 300 11 700 truth > $out/fuzz_truth_seed11.txt 2>&1; rc1=$?; tail -3 $out/fuzz_truth_seed11.txt
FUZZ_NOISE_LO=1e-6 timeout -k 10 420 python tools/fuzz_parity.py 300 12 700 truth > $out/fuzz_truth_lownoise_seed12.txt 2>&1; rc2=$?; tail -3 $out/fuzz_truth_lownoise_seed12.txt
timeout -k 10 420 python tools/fuzz_parity.py 300 2 7400 > $out/fuzz_parity_n7400_seed2.txt 2>&1; rc3=$?; tail -1 $out/fuzz_parity_n7400_seed2.txt
exit $(( rc1 + rc2 + rc3 ))
