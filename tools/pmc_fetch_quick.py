"""FETCH_SIZE (KB, raw) of one counter CSV: the sweep of the last evaluation and the predvar launch -> GB after the x2 of the guide"""
import csv, sys
rows = {}
order = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if r["Counter_Name"] != "FETCH_SIZE":
            continue
        d = int(r["Dispatch_Id"])
        if d not in rows:
            rows[d] = [r["Kernel_Name"], 0.0]
            order.append(d)
        rows[d][1] += float(r["Counter_Value"])
seq = [rows[d] for d in sorted(order)]
kb = [i for i, (k, _) in enumerate(seq) if "kbuild" in k]
sweep = sum(v for k, v in seq[kb[-2]:kb[-1]] if ("gemm_nt" in k or "leaf" in k))
pv = [v for k, v in seq if "predvar" in k]
print("sweep fetch %.2f GB (x2: %.2f GB)   predvar fetch %.2f GB (x2: %.2f GB)" % (sweep / 1e6 * 1.024, 2 * sweep / 1e6 * 1.024, (pv[-1] if pv else 0) / 1e6 * 1.024, 2 * (pv[-1] if pv else 0) / 1e6 * 1.024))
