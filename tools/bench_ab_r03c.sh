#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-bab3c}; mkdir -p $out
b() { printf "%-44s" "$1"; env $2 timeout -k 10 300 python bench.py --concurrency 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['wall_ms_per_evaluation'], d['roofline']['achieved'], d['roofline']['uncontended']['avg_launch_ms'])"; }
{
b "default" "A=1"
b "BULK_EVERY=2" "MFGP_BULK_EVERY=2"
b "BULK_EVERY=4" "MFGP_BULK_EVERY=4"
b "SHIFT=1" "MFGP_SHIFT=1"
b "XPANEL_MERGE=0" "MFGP_XPANEL_MERGE=0"
b "KINV_STREAM=0" "MFGP_KINV_STREAM=0"
b "CHAIN_TILE=32" "MFGP_CHAIN_TILE=32"
b "CHAIN_SLIM=0" "MFGP_CHAIN_SLIM=0"
b "default (repeat)" "A=1"
} | tee $out/bench_ab.txt
