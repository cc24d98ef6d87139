#!/bin/bash
# What would a leaf that fits beside ONE bulk workgroup buy?  Timing placebo (results garbage) of 141312 / 97280 / 30720 B.
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-pl}; out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
# lab build: the placebo exists only under -DMFGP_LAB_PLACEBO=1 (this box's copy of the tree is thrown away afterwards)
MFGP_BUILD_DEFINES="-DMFGP_LAB_PLACEBO=1" python -m multifidelity_datafusion_gps_amd.build --force > $out/lab_build.log 2>&1 || exit 1
for cfg in "0 8 0" "141312 8 0" "97280 8 0" "97280 4 0" "97280 4 1" "30720 8 0" "30720 4 0" "30720 4 1"; do
  set -- $cfg
  echo "LEAF_PLACEBO=$1 CHAIN_WAVES=$2 COLS_STREAM=$3"; MFGP_LEAF_PLACEBO=$1 MFGP_CHAIN_WAVES=$2 MFGP_COLS_STREAM=$3 python tools/time_eval.py 4096 6144 8192 16384 2>&1 | cut -c1-100
done | tee $out/time_eval_ab.txt
cd /tmp
n=8192
for cfg in "97280 4 0" "30720 4 0"; do
  set -- $cfg
  MFGP_LEAF_PLACEBO=$1 MFGP_CHAIN_WAVES=$2 MFGP_COLS_STREAM=$3 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$n -- python3 $GRAFT_REPO_ROOT/tools/time_eval.py $n > $out/trace_$n.log 2>&1
  (cd $GRAFT_REPO_ROOT && python tools/chain_account.py $out/trace_$n > $out/chain_account_${n}_$1.txt 2>&1)
  echo "placebo $1 waves $2:"; cat $out/chain_account_${n}_$1.txt | head -3
  rm -rf $out/trace_$n
done
