#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-bab3b}; mkdir -p $out
b() { printf "%-44s" "$1"; env $2 timeout -k 10 300 python bench.py --concurrency $3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['wall_ms_per_evaluation'], d['roofline']['achieved'], d['roofline']['uncontended']['avg_launch_ms'])"; }
{
b "default (MACRO 4) c2" "A=1" 2
b "MACRO=5 c2" "MFGP_MACRO=5" 2
b "MACRO=6 c2" "MFGP_MACRO=6" 2
b "MACRO=8 c2" "MFGP_MACRO=8" 2
b "MACRO=3 c2" "MFGP_MACRO=3" 2
b "T128_MIN=300 c2" "MFGP_T128_MIN=300" 2
b "T128_MIN=1000 c2" "MFGP_T128_MIN=1000" 2
b "default c3" "A=1" 3
b "default c1" "A=1" 1
b "default c2 (repeat)" "A=1" 2
} | tee $out/bench_ab.txt
