#!/bin/bash
set -o pipefail
out=gpurun_out/r04l; mkdir -p $out
python -m pytest tests/test_gpu_multirank.py -m gpu -x -q -k "group_of_one" > $out/tests1.log 2>&1 || { tail -40 $out/tests1.log; exit 1; }
tail -2 $out/tests1.log
python tools/shard_projection.py 4096 8192 > $out/shard_projection.txt 2>&1; cat $out/shard_projection.txt
timeout -k 10 600 python -m pytest tests/test_gpu_multirank.py -m gpu -x -q -k "rccl_communicator" > $out/tests2.log 2>&1 || { tail -60 $out/tests2.log; exit 1; }
tail -2 $out/tests2.log
