#!/bin/bash
set -o pipefail
out=gpurun_out/r04m; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -60 $out/tests.log; exit 1; }
tail -2 $out/tests.log
timeout -k 10 300 python bench.py --gpus 2 --single-device --no-cpu-baseline --points 2048 --steps 1 --warmup 1 > $out/bench_n2_rehearsal.json 2> $out/bench_n2.err || { tail -20 $out/bench_n2.err; exit 1; }
python -c "
import json; d=json.load(open('$out/bench_n2_rehearsal.json')); print('2-rank rehearsal N=2048:', d['value'], d['config']['rccl_ranks'], d['config']['collectives'][:60])"
timeout -k 10 300 python bench.py --gpus 3 --single-device --no-cpu-baseline --points 2048 --steps 1 --warmup 1 --restarts 2 > $out/bench_n3_rehearsal.json 2> $out/bench_n3.err || { tail -20 $out/bench_n3.err; exit 1; }
python -c "
import json; d=json.load(open('$out/bench_n3_rehearsal.json')); print('3-rank rehearsal N=2048, 2 restarts:', d['value'], d['config']['rccl_ranks'])"
