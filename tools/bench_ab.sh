#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-bab}; mkdir -p $out
b() { printf "%-36s" "$1"; env $2 timeout -k 10 300 python bench.py --concurrency $3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['wall_ms_per_evaluation'], d['roofline']['achieved'], d['roofline']['uncontended']['avg_launch_ms'])"; }
{
b "default c2" "A=1" 2
b "CHAIN_SLIM=0 c2" "MFGP_CHAIN_SLIM=0" 2
b "CHAIN_SLIM=0 T128_MIN=300 c2" "MFGP_CHAIN_SLIM=0 MFGP_T128_MIN=300" 2
b "SHIFT=1 c2" "MFGP_SHIFT=1" 2
b "XPANEL_MERGE=0 c2" "MFGP_XPANEL_MERGE=0" 2
b "BULK_EVERY=2 c2" "MFGP_BULK_EVERY=2" 2
b "default c2 (repeat)" "A=1" 2
} | tee $out/bench_ab.txt
