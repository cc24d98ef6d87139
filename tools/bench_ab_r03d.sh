#!/bin/bash
# bench A/B under the 2-hardware-queue default (second session of round 3)
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-bab3d}; mkdir -p $out
b() { printf "%-44s" "$1"; env $2 timeout -k 10 300 python bench.py --concurrency ${3:-2} --no-cpu-baseline --no-power 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['wall_ms_per_evaluation'], d['roofline']['achieved'], d['roofline']['uncontended']['avg_launch_ms'])"; }
{
b "default c2" "A=1"
b "c3" "A=1" 3
b "c4" "A=1" 4
b "MACRO=6 c2" "MFGP_MACRO=6"
b "MACRO=8 c2" "MFGP_MACRO=8"
b "MACRO=3 c2" "MFGP_MACRO=3"
b "BULK_EVERY=2 c2" "MFGP_BULK_EVERY=2"
b "CHAIN_WAVES=4 c2" "MFGP_CHAIN_WAVES=4"
b "COLS_STREAM=1 c2" "MFGP_COLS_STREAM=1"
b "U_RESERVE=1 c2" "MFGP_U_RESERVE=1"
b "HW_QUEUES=4 c2" "GPU_MAX_HW_QUEUES=4"
b "default c2 (repeat)" "A=1"
} | tee $out/bench_ab.txt
