#!/bin/bash
set -o pipefail
out=gpurun_out/r04i; mkdir -p $out
python -m pytest tests/test_gpu_models.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
python tools/midsize_fit.py 256 1024 2048 4096 > $out/midsize_fit.txt 2>&1; cat $out/midsize_fit.txt
python bench.py --no-cpu-baseline --no-power > $out/bench_lockstep.json 2> $out/bench_lockstep.err; python -c "
import json; d=json.load(open('$out/bench_lockstep.json')); print('lockstep 2 lanes:', d['value'], d['roofline']['achieved'], d['config']['evals_issued_rank0_per_step'], d['serial_floor'])"
python bench.py --no-cpu-baseline --no-power --lanes 1 > $out/bench_lockstep_1lane.json 2>> $out/bench_lockstep.err; python -c "
import json; d=json.load(open('$out/bench_lockstep_1lane.json')); print('lockstep 1 lane:', d['value'], d['roofline']['achieved'])"
python bench.py --no-cpu-baseline --no-power --lockstep 0 > $out/bench_conc.json 2>> $out/bench_lockstep.err; python -c "
import json; d=json.load(open('$out/bench_conc.json')); print('round-3 concurrent:', d['value'], d['roofline']['achieved'], d['result_checksum'])"
python -c "
import json; d=json.load(open('$out/bench_lockstep.json')); print(d['result_checksum'])"
tail -5 $out/bench_lockstep.err
