#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-ab}; mkdir -p $out
run() { printf "%-44s" "$1"; env $2 timeout -k 10 120 python tools/time_eval.py $3 2>&1 | awk '{printf "  %s %s", $1, $3}' ; echo; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -m gpu -x -q > $out/tests.log 2>&1; tail -3 $out/tests.log
{
run "default" "A=1" "1024 2048 4096 6144 8192 16384"
run "T128_MIN=600" "MFGP_T128_MIN=600" "4096 6144 8192 16384"
run "T128_MIN=1000" "MFGP_T128_MIN=1000" "4096 8192"
run "MACRO=6" "MFGP_MACRO=6" "6144 8192"
run "MACRO=4" "MFGP_MACRO=4" "6144 8192"
run "MACRO=4 BULK_EVERY=2" "MFGP_MACRO=4 MFGP_BULK_EVERY=2" "6144 8192"
run "MACRO=8 (6144)" "MFGP_MACRO=8" "6144"
run "PLAN=levels" "MFGP_PLAN=levels" "4096 8192"
} | tee $out/plan_ab2.txt
for c in 1 2 3; do
echo "== bench concurrency $c"
timeout -k 10 400 python bench.py --concurrency $c --no-cpu-baseline > $out/bench_c$c.json 2> $out/bench.err; python -c "
import json,sys; d=json.loads(open('$out/bench_c$c.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['wall_ms_per_evaluation'], d['roofline']['achieved'], d['roofline']['uncontended']['avg_launch_ms'])"
done
echo "== bench MFGP_PLAN=levels concurrency 2"
MFGP_PLAN=levels timeout -k 10 400 python bench.py --concurrency 2 --no-cpu-baseline > $out/bench_levels.json 2> $out/bench.err; python -c "
import json,sys; d=json.loads(open('$out/bench_levels.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['wall_ms_per_evaluation'])"
