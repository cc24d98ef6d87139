#!/bin/bash
# round 3: macro-panel length vs size, one evaluation alone (tools/time_eval.py total ms), after the bulk kernels got faster
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r03macro}; mkdir -p $out
{
for m in 2 3 4 5 6 8 12 16; do
  printf "MACRO=%-3s" $m
  case $m in
    2|3) sizes="2048 3072 4096 5120 6144 7168 8192";;
    4|5|6) sizes="4096 5120 6144 7168 8192 10240 12288";;
    *) sizes="8192 10240 12288 16384";;
  esac
  MFGP_MACRO=$m timeout -k 10 200 python tools/time_eval.py $sizes 2>&1 | awk '{printf "  %s %s", $1, $3}'; echo
done
} | tee $out/macro_sweep.txt
