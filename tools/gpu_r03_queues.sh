#!/bin/bash
# HIP streams vs hardware queues: GPU_MAX_HW_QUEUES (ROCm's default is 4 per process; a bench rank holds 4 engine handles x 2
# streams) -- the bench and cfg3's HF level (chain-bound N = 4096 with three evaluations in flight)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-hq}; out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
for q in ${QUEUE_LIST:-default 8 12 2}; do
  if [ "$q" = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  printf "GPU_MAX_HW_QUEUES=$q bench "; timeout -k 10 300 python bench.py --no-cpu-baseline --no-power 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['wall_ms_per_evaluation'], d['roofline']['achieved'], d['roofline']['uncontended']['avg_launch_ms'])"
  printf "GPU_MAX_HW_QUEUES=$q "; timeout -k 10 300 python tools/run_configs.py 2>&1 | grep "^cfg3" | cut -c1-160
done | tee $out/queues.txt
