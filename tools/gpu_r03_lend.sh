#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-lend}; mkdir -p $out
b() { printf "%-44s" "$1"; timeout -k 10 300 python bench.py $2 --no-cpu-baseline --no-power 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['wall_ms_per_evaluation'], d['roofline']['achieved'], d['result_checksum']['mean_sum'])"; }
{
b "default (2 aux)" ""
b "1 aux (other lane) + lend main" "--aux 1 --lend-main 1 --aux-order reversed"
b "1 aux (other lane), no lend" "--aux 1 --aux-order reversed"
b "2 aux reversed" "--aux-order reversed"
b "2 aux reversed + lend" "--aux-order reversed --lend-main 1"
b "default (2 aux)" ""
b "1 aux (other lane) + lend main" "--aux 1 --lend-main 1 --aux-order reversed"
} | tee $out/lend.txt
