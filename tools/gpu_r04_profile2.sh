#!/bin/bash
# second call: the un-profiled bench lines (power sampled beside the first; the default line with the CPU baseline last)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r04p}; out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
timeout -k 10 300 python3 bench.py --no-cpu-baseline --power > $out/bench_power.json 2> $out/bench_power.err; tail -c 1500 $out/bench_power.json
timeout -k 10 300 python3 bench.py --no-cpu-baseline --lockstep 0 > $out/bench_concurrent_r3_mode.json 2>> $out/bench_power.err; python3 -c "
import json; d=json.loads([l for l in open('$out/bench_concurrent_r3_mode.json') if l.startswith('{')][-1]); print('round-3 mode (concurrent restarts):', d['value'], d['roofline']['achieved'])"
timeout -k 10 800 python3 bench.py > $out/bench.json 2> $out/bench.err; tail -c 3000 $out/bench.json
