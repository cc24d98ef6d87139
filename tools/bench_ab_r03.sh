#!/bin/bash
# round 3 A/B of the bench workload (N = 8192, three evaluations in flight) over planner switches, after the bulk kernels
# moved to the 4x4x4 / LDS-DMA body: value ms | wall ms per evaluation | aggregate sweep TFLOP/s | uncontended sweep ms
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-bab3}; mkdir -p $out
b() { printf "%-44s" "$1"; env $2 timeout -k 10 300 python bench.py --concurrency $3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['wall_ms_per_evaluation'], d['roofline']['achieved'], d['roofline']['uncontended']['avg_launch_ms'])"; }
{
b "default c2" "A=1" 2
b "MACRO=12 c2" "MFGP_MACRO=12" 2
b "MACRO=16 c2" "MFGP_MACRO=16" 2
b "MACRO=6 c2" "MFGP_MACRO=6" 2
b "T128_MIN=300 c2" "MFGP_T128_MIN=300" 2
b "T128_MIN=100 c2" "MFGP_T128_MIN=100" 2
b "T128_MIN=1 KINV_T128_MIN=1 c2" "MFGP_T128_MIN=1 MFGP_KINV_T128_MIN=1" 2
b "MACRO=16 T128_MIN=100 c2" "MFGP_MACRO=16 MFGP_T128_MIN=100" 2
b "default c3" "A=1" 3
b "default c4" "A=1" 4
b "CHAIN_SLIM=0 c2" "MFGP_CHAIN_SLIM=0" 2
b "default c2 (repeat)" "A=1" 2
} | tee $out/bench_ab.txt
