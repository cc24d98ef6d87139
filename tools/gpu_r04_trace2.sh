#!/bin/bash
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r04f; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for m in 2 4 8; do
  MFGP_MACRO=$m timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace_4096_6_m$m -- python3 $GRAFT_REPO_ROOT/tools/batch_trace.py 4096 6 4 > $out/trace_m$m.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/trace_last_pass.py $out/trace_4096_6_m$m 2000 > $out/timeline_4096_6_m$m.txt 2>&1
  head -8 $out/timeline_4096_6_m$m.txt
  K=$((m*128)); grep "t128" $out/timeline_4096_6_m$m.txt | grep -v "#" | awk -v K=$K '{b=$4; d=$(NF-1); tf=b*2*128*128*K/d/1e6; printf "%s blocks %d dur %.1f us -> %.1f TF\n", $1, b, d, tf}' | head -20
  find $out/trace_4096_6_m$m -name "*.csv" -size +5M -delete
done
rocm-smi --showpower --showclocks 2>/dev/null | head -20
