#!/bin/bash
# runtime knobs that touch launch latency: one evaluation alone by size, and the bench
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-knobs}; mkdir -p $out
for kv in "A=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "AMD_DIRECT_DISPATCH=0" "HSA_ENABLE_INTERRUPT=0" "ROC_ACTIVE_WAIT_TIMEOUT=1000000" "A=1"; do
  echo "== $kv"
  env $kv python tools/time_eval.py 128 1024 2048 4096 8192 2>&1 | cut -c1-28 | tr '\n' ' '; echo
  printf "bench "; env $kv timeout -k 10 300 python bench.py --no-cpu-baseline --no-power 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['achieved'], d['roofline']['uncontended']['avg_launch_ms'])"
done | tee $out/knobs.txt
