#!/bin/bash
# round 4: does operand traffic cost time on the 128-tile kernel?  (tools/gemm_lab/order_traffic.py)
out=$GRAFT_REPO_ROOT/gpurun_out/r04o; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 400 python3 $R/tools/gemm_lab/order_traffic.py time > $out/order_time.txt 2>&1 || { tail -5 $out/order_time.txt; exit 1; }
cat $out/order_time.txt
pmc() {  # name K order bi bj
  timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_$1 -- python3 $R/tools/gemm_lab/order_traffic.py one $2 $3 $4 $5 > $out/pmc_$1.log 2>&1 || { tail -5 $out/pmc_$1.log; return 1; }
  f=$(find $out/pmc_$1 -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$1" <<'PY'
import csv, sys
tot = {}
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE" and "lab_fill" not in r["Kernel_Name"]:
        tot[int(r["Dispatch_Id"])] = tot.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
last = tot[max(tot)]
print("%-16s FETCH_SIZE %.3f GB raw (x2 per the guide: %.3f GB)" % (sys.argv[2], last * 1024 / 1e9, 2 * last * 1024 / 1e9))
PY
  rm -rf $out/pmc_$1
}
{
for K in 512 2048; do
  pmc K${K}_rowmajor $K 0 0 0 && pmc K${K}_same $K 1 0 0 && pmc K${K}_1x8 $K 2 1 8 && pmc K${K}_4x8 $K 2 4 8 && pmc K${K}_8x8 $K 2 8 8 && pmc K${K}_16x16 $K 2 16 16 || exit 1
done
} | tee $out/order_fetch.txt
