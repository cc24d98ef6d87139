#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab}; mkdir -p $out
export TMPDIR=/tmp
run() { printf "%-44s" "$1"; env $2 timeout -k 10 120 python tools/time_eval.py $3 2>&1 | awk '{printf "  %s %s", $1, $3}' ; echo; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; tail -3 $out/tests.log
{
run "default" "A=1" "128 1024 2048 4096 6144 8192 16384"
run "MACRO=8" "MFGP_MACRO=8" "6144 8192"
} | tee $out/plan_ab2.txt
for c in 1 2 3; do
echo "== bench concurrency $c"
timeout -k 10 400 python bench.py --concurrency $c --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python -c "
import json,sys; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['wall_ms_per_evaluation'], d['roofline']['achieved'], d['roofline']['uncontended']['avg_launch_ms'])"
done
