#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-bab}; mkdir -p $out
b() { printf "%-36s" "$1"; env $2 timeout -k 10 300 python bench.py --concurrency $3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['wall_ms_per_evaluation'], d['roofline']['achieved'], d['roofline']['uncontended']['avg_launch_ms'])"; }
{
b "default c2" "A=1" 2
b "MACRO=8 c2" "MFGP_MACRO=8" 2
b "MACRO=12 c2" "MFGP_MACRO=12" 2
b "MACRO=16 c2" "MFGP_MACRO=16" 2
b "MACRO=8 c3" "MFGP_MACRO=8" 3
b "MACRO=12 c3" "MFGP_MACRO=12" 3
b "MACRO=8 T128_MIN=300 c2" "MFGP_MACRO=8 MFGP_T128_MIN=300" 2
b "MACRO=4 c2" "MFGP_MACRO=4" 2
b "default c2 (repeat)" "A=1" 2
} | tee $out/bench_ab.txt
