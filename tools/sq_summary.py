"""Per-kernel SQ counter fractions from a rocprofv3 --pmc pass (counter_collection.csv):

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
              SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d DIR -- python3 tools/predv_once.py 8192
    python tools/sq_summary.py DIR [name filter]

wait_any / wait_inst / active / valu / lds are fractions of SQ_WAVE_CYCLES; mfma = SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES)."""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
traces = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void mfgp::", "")
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k].add(row["Dispatch_Id"])
dur = defaultdict(list)
for f in traces:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void mfgp::", "")
        dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k, c in sorted(acc.items()):
    if flt not in k: continue
    w = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    cu = c.get("SQ_BUSY_CU_CYCLES", 0.0) or 1.0
    us = sum(dur[k]) / len(dur[k]) if dur[k] else float("nan")
    print("%-46s n=%3d avg %7.1f us wait_any %.2f wait_inst %.2f active %.2f | valu %.2f lds %.2f | mfma_busy/(4*cu_busy) %.3f"
          % (k[:46], len(cnt[k]), us, c.get("SQ_WAIT_ANY", 0) / w, c.get("SQ_WAIT_INST_ANY", 0) / w, c.get("SQ_ACTIVE_INST_ANY", 0) / w,
             c.get("SQ_ACTIVE_INST_VALU", 0) / w, c.get("SQ_ACTIVE_INST_LDS", 0) / w, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * cu)))
