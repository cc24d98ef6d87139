#!/bin/bash
# copies what `tools/gpu_session.sh profile <out>` left under gpurun_out/<out>/ into profiles/ under the round's names
#   bash tools/copy_profile_records.sh r06_profile r06
set -e
S=gpurun_out/${1:?session directory}; P=profiles/${2:?prefix}
cp $S/adapt_kernel_stats.csv ${P}_adapt_kernel_stats.csv
cp $S/bench_kernel_stats.csv ${P}_bench_kernel_stats.csv
cp $S/bench_profiled.json ${P}_bench_n1_profiled.json
cp $S/mfma_counters.json ${P}_mfma_counters.json
cp $S/pmc.json ${P}_pmc.json
for n in 4096 8192; do cp $S/plan_flops_$n.txt ${P}_plan_flops_$n.txt; cp $S/timeline_$n.txt ${P}_timeline_$n.txt; cp $S/timeline_batch_${n}_B4.txt ${P}_timeline_batch_${n}_B4.txt; done
for c in FETCH WRITE; do l=$(echo $c | tr A-Z a-z); for w in predv_once time_eval; do cp $S/pmc_${c}_SIZE_${w}_8192.csv ${P}_pmc_${l}_size_${w}_8192.csv; done; done
python3 - "$S" "$P" <<'PY'
import json, re, sys
S, P = sys.argv[1], sys.argv[2]
h = json.load(open(P + "_pmc.json"))["csrc_hash"]
assert json.load(open(P + "_mfma_counters.json")).get("csrc_hash") == h
old = open(P + "_adapt_sq.txt").read()
new = open(S + "/adapt_sq.txt").read()
i = old.index("# ---- the same kernel with v_mfma_f64_4x4x4_4b")
hdr = re.sub(r"build [0-9a-f]{16}", "build " + h, old[:old.index("__amd_rocclr_copyBuffer")], count=1)
open(P + "_adapt_sq.txt", "w").write(hdr + new + "\n" + old[i:])
print("records of library build", h)
PY
