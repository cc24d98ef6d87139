#!/bin/bash
# HBM-side traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) and time of one evaluation + one predict at
# N = 8192 by super-block shape of the tile enumeration: the bulk launches of the sweep (MFGP_BULK_BI x MFGP_BULK_BJ) and the
# predictive-variance launch (MFGP_PREDV_BI x MFGP_PREDV_BR).  usage: gpu_r03_traffic.sh <tag> "bi bj pi pr" ...
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for cfg in "$@"; do
  set -- $cfg
  export MFGP_BULK_BI=$1 MFGP_BULK_BJ=$2 MFGP_PREDV_BI=$3 MFGP_PREDV_BR=$4
  name=b${1}x${2}_p${3}x${4}
  t=$(python3 $R/tools/time_eval.py 8192 | grep -o 'total [0-9.]* ms\|var [0-9.]* ([0-9.]* TF)' | tr '\n' ' ')
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python3 $R/tools/time_eval.py 8192 > $out/pmc_$c.log 2>&1
    f=$(find $out/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/${name}_$c.csv
    rm -rf $out/pmc_$c
  done
  (cd $R && python3 tools/pmc_summary.py 8192 $out/$name.json FETCH_SIZE=$out/${name}_FETCH_SIZE.csv WRITE_SIZE=$out/${name}_WRITE_SIZE.csv > /dev/null 2>&1
   python3 -c "
import json; d=json.load(open('$out/$name.json'))
print('bulk %sx%s predv %sx%s | %s | sweep traffic %.2f GB (fetch raw %.2f GB, write %.2f GB) | predvar traffic %.2f GB' % ('$1','$2','$3','$4','$t', d['sweep']['traffic_bytes']/1e9, d['sweep']['fetch_kb_raw']*1024/1e9, d['sweep']['write_kb']*1024/1e9, d['mfgp_predvar_f64']['traffic_bytes']/1e9))")
  rm -f $out/${name}_FETCH_SIZE.csv $out/${name}_WRITE_SIZE.csv
done | tee $out/traffic_ab.txt
