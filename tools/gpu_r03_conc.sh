#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-conc}; mkdir -p $out
for c in 2 3 4 5 2 3; do
  printf "concurrency=$c bench "; timeout -k 10 300 python bench.py --concurrency $c --no-cpu-baseline --no-power 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['wall_ms_per_evaluation'], d['roofline']['achieved'])"
  printf "concurrency=$c "; MFGP_RESTART_CONC=$c timeout -k 10 300 python tools/run_configs.py 2>&1 | grep "^cfg3\|^cfg4" | sed 's/(budget 20 evals\/run)//' | cut -c1-130 | tr '\n' '|'; echo
done | tee $out/conc.txt
