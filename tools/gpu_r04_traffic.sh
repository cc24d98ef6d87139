#!/bin/bash
# round 4: FETCH_SIZE of the sweep (by macro panel length / super-block shape) and of the predictive-variance launch (by super-block shape)
out=$GRAFT_REPO_ROOT/gpurun_out/r04k; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
run() {  # name, env...
  name=$1; shift
  env "$@" python3 $R/tools/time_eval.py 8192 > $out/time_$name.txt 2>&1
  for v in "$@"; do export $v; done
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_$name -- python3 $R/tools/time_eval.py 8192 > $out/pmc_$name.log 2>&1
  for v in "$@"; do unset ${v%%=*}; done
  f=$(find $out/pmc_$name -name "*counter_collection.csv" | head -1)
  printf "%-34s %s | %s\n" "$name" "$(python3 $R/tools/pmc_fetch_quick.py $f)" "$(tail -1 $out/time_$name.txt | cut -c1-150)"
  rm -rf $out/pmc_$name
}
run default MFGP_X_DUMMY=0
run macro8 MFGP_MACRO=8
run macro16 MFGP_MACRO=16
run bulk8x4 MFGP_X_BULK_BI=8 MFGP_X_BULK_BJ=4
run bulk8x8 MFGP_X_BULK_BI=8 MFGP_X_BULK_BJ=8
run macro8_bulk8x8 MFGP_MACRO=8 MFGP_X_BULK_BI=8 MFGP_X_BULK_BJ=8
run predv8x8 MFGP_X_PREDV_BI=8 MFGP_X_PREDV_BR=8
run predv4x8 MFGP_X_PREDV_BI=4 MFGP_X_PREDV_BR=8
run predv16x4 MFGP_X_PREDV_BI=16 MFGP_X_PREDV_BR=4
run predv4x16 MFGP_X_PREDV_BI=4 MFGP_X_PREDV_BR=16
