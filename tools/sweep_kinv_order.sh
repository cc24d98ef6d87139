# K^-1 launch: time and HBM fetch per super-block shape of the XCD-aware task order (GPU box)
export TMPDIR=/tmp
for cfg in "1 8" "2 8" "4 4" "2 16" "4 8" "3 8" "1 16"; do
  set -- $cfg
  t=$(MFGP_KINV_BI=$1 MFGP_KINV_BJ=$2 timeout -k 10 100 python tools/time_eval.py 8192 | grep -o 'kinv [0-9.]* ([0-9.]* TF)')
  rm -rf gpurun_out/pmc_sweep
  MFGP_KINV_BI=$1 MFGP_KINV_BJ=$2 timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_sweep -- python3 tools/time_eval.py 8192 > /dev/null 2>&1
  f=$(python3 - <<'PY'
import csv, glob, statistics
p = glob.glob("gpurun_out/pmc_sweep/*/*counter_collection.csv")[0]
v = {}
for r in csv.DictReader(open(p)):
    if "kinv_syrk_f64" in r["Kernel_Name"] and "t64" not in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        v[r["Dispatch_Id"]] = v.get(r["Dispatch_Id"], 0) + float(r["Counter_Value"])
print("fetch raw %.2f GB" % (statistics.median(v.values()) / 1e6))
PY
)
  echo "BI=$1 BJ=$2: $t  $f"
done
