#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-ab}; mkdir -p $out
run() { printf "%-44s" "$1"; env $2 timeout -k 10 120 python tools/time_eval.py $3 2>&1 | awk '{printf "  %s %s", $1, $3}' ; echo; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; tail -5 $out/tests.log
MFGP_LEAF_STAMPS=1 timeout -k 10 60 python tools/leaf_stamps.py 2>&1 | tail -3
MFGP_LEAF=2 MFGP_LEAF_STAMPS=1 timeout -k 10 60 python tools/leaf_stamps.py 2>&1 | tail -3
{
run "default" "A=1" "128 512 1024 2048 4096 6144 8192 16384"
run "LEAF=2 (two-phase leaf)" "MFGP_LEAF=2" "128 512 1024 2048 4096 8192"
run "KBUILD_FAST=0" "MFGP_KBUILD_FAST=0" "4096 8192"
} | tee $out/plan_ab2.txt
timeout -k 10 100 python tools/time_eval.py 8192 > $out/te8192.txt 2>&1; cat $out/te8192.txt
MFGP_KBUILD_FAST=0 timeout -k 10 100 python tools/time_eval.py 8192 2>&1
for c in 2; do
echo "== bench concurrency $c"
timeout -k 10 400 python bench.py --concurrency $c --no-cpu-baseline > $out/bench_c$c.json 2> $out/bench.err; python -c "
import json,sys; d=json.loads(open('$out/bench_c$c.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['wall_ms_per_evaluation'], d['roofline']['achieved'], d['roofline']['uncontended']['avg_launch_ms'], d['roofline_kbuild'])"
done
