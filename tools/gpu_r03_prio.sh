#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-prio}; mkdir -p $out
b() { printf "%-50s" "$1"; env $2 timeout -k 10 300 python bench.py --no-cpu-baseline --no-power 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['wall_ms_per_evaluation'], d['roofline']['achieved'], d['roofline']['uncontended']['avg_launch_ms'])"
  printf "%-50s" ""; env $2 timeout -k 10 300 python tools/run_configs.py 2>&1 | grep "^cfg3\|^cfg4" | sed 's/(budget 20 evals\/run)//' | cut -c1-130 | tr '\n' '|'; echo; }
{
b "chain hi / bulk lo, Q=2 (default)" "A=1"
b "chain hi / bulk normal, Q=2" "MFGP_PRIO_BULK=normal"
b "chain normal / bulk lo, Q=2" "MFGP_PRIO_CHAIN=normal"
b "chain normal / bulk normal, Q=2" "MFGP_PRIO_CHAIN=normal MFGP_PRIO_BULK=normal"
b "chain normal / bulk normal, Q=4" "MFGP_PRIO_CHAIN=normal MFGP_PRIO_BULK=normal GPU_MAX_HW_QUEUES=4"
b "chain normal / bulk normal, Q=3" "MFGP_PRIO_CHAIN=normal MFGP_PRIO_BULK=normal GPU_MAX_HW_QUEUES=3"
b "chain hi / bulk normal, Q=4" "MFGP_PRIO_BULK=normal GPU_MAX_HW_QUEUES=4"
} | tee $out/prio.txt
