#!/bin/bash
# round 4, second session: lock step by run generators (one loop per lane) against a thread per run
set -o pipefail
out=gpurun_out/${1:-r04ls}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_multirank.py tests/test_gpu_reference_l3.py -x -q > $out/tests.log 2>&1; rc=$?; tail -5 $out/tests.log
[ $rc = 0 ] || exit $rc
python tools/midsize_fit.py 256 512 1024 2048 4096 > $out/midsize_fit.txt 2>&1; cat $out/midsize_fit.txt
python tools/midsize_fit.py --evals 20 2048 4096 >> $out/midsize_fit.txt 2>&1; tail -3 $out/midsize_fit.txt
