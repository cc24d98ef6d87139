#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab}; mkdir -p $out
export TMPDIR=/tmp
run() { printf "%-44s" "$1"; env $2 timeout -k 10 120 python tools/time_eval.py $3 2>&1 | awk '{printf "  %s %s", $1, $3}' ; echo; }
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -x -q > $out/tests.log 2>&1; tail -3 $out/tests.log
MFGP_LEAF_STAMPS=1 timeout -k 10 60 python tools/leaf_stamps.py 2>&1 | tail -2
{
run "default (4x4 super-blocks, XCD deal)" "A=1" "128 1024 2048 4096 6144 8192 16384"
run "BULK_XCD=0" "MFGP_BULK_XCD=0" "4096 6144 8192 16384"
run "BULK_BI=2 BULK_BJ=8" "MFGP_BULK_BI=2 MFGP_BULK_BJ=8" "4096 8192"
run "BULK_BI=8 BULK_BJ=4" "MFGP_BULK_BI=8 MFGP_BULK_BJ=4" "4096 8192"
run "BULK_BI=1 BULK_BJ=8" "MFGP_BULK_BI=1 MFGP_BULK_BJ=8" "8192"
run "BULK_BI=4 BULK_BJ=8" "MFGP_BULK_BI=4 MFGP_BULK_BJ=8" "8192 16384"
} | tee $out/plan_ab2.txt
cd /tmp
for v in 1 0; do
  MFGP_BULK_XCD=$v timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch_xcd$v -- python3 $GRAFT_REPO_ROOT/tools/time_eval.py 8192 > $out/pmc_fetch_xcd$v.log 2>&1
  f=$(find $out/pmc_fetch_xcd$v -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys
rows={}
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"]!="FETCH_SIZE": continue
    d=int(r["Dispatch_Id"]); rows.setdefault(d,[r["Kernel_Name"],0.0]); rows[d][1]+=float(r["Counter_Value"])
ds=sorted(rows); kb=[i for i,d in enumerate(ds) if "kbuild" in rows[d][0]]
s=sum(rows[d][1] for d in ds[kb[-2]:kb[-1]] if "gemm" in rows[d][0] or "leaf" in rows[d][0])
print("sweep FETCH_SIZE raw %.2f GB (x2 = %.2f GB)"%(s/1e6, 2*s/1e6))
PY
  find $out/pmc_fetch_xcd$v -name "*.csv" -size +1M -delete
done
