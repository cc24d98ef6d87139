#!/bin/bash
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r04c; mkdir -p $out
export TMPDIR=/tmp
for n in 1500 2100 4096; do python tools/plan_bitwise.py $n; done > $out/plan_bitwise.txt 2>&1; cat $out/plan_bitwise.txt
cd /tmp
for nb in "2048 4" "4096 6" "4096 1"; do set -- $nb
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$1_$2 -- python3 $GRAFT_REPO_ROOT/tools/batch_trace.py $1 $2 > $out/trace_$1_$2.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/trace_last_pass.py $out/trace_$1_$2 600 > $out/timeline_$1_$2.txt 2>&1
  head -12 $out/timeline_$1_$2.txt
  find $out/trace_$1_$2 -name "*.csv" -size +20M -delete
done
