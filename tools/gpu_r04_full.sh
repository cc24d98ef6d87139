#!/bin/bash
# the round's acceptance run: the whole -m gpu suite, smoke(), the default bench line (with the CPU baseline)
set -o pipefail
out=gpurun_out/${1:-r04full}; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=10 > $out/gputests.log 2>&1; rc=$?; tail -15 $out/gputests.log; cp gpurun_out/parity_errors.json $out/ 2>/dev/null
[ $rc = 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
tail -1 $out/smoke.log
timeout -k 10 800 python bench.py > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python -c "
import json; d=json.loads([l for l in open('$out/bench.json') if l.startswith('{')][-1])
print('bench', d['value'], 'roofline', d['roofline']['achieved'], d['roofline']['frac'], 'traffic', d['roofline']['traffic'], d['roofline']['traffic_source'][:60])
cb=d['cpu_baseline']; print('cpu', cb['value'], cb['cores'], cb['lf_eval_s'], cb['hf_eval_s'], cb['warmed_up'], cb['measured_s'], cb['lapack_share_of_one_hf_eval'], cb['blas_threads_tried_gflops'])
print('floor', d['serial_floor_ms'], d.get('serial_floor_sharded_projection'))"
