"""Summarise rocprofv3 --pmc counter_collection CSVs (one pass per counter) into per-kernel, per-launch HBM traffic.
usage: pmc_summary.py N out.json FETCH_SIZE=<csv> WRITE_SIZE=<csv>
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
prescribes for wide coalesced reads on gfx950; the median over a kernel's launches is reported."""
import csv, json, sys, statistics

KERNELS = ["mfgp_kinv_syrk_f64", "mfgp_kbuild_f64<0>", "mfgp_predvar_f64", "mfgp_grad_tiles_f64", "mfgp_predv_skinny_f64"]


def per_kernel(path, counter):
    acc = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            for k in KERNELS:
                if name.startswith(k) or ("mfgp::" + k) in name or k in name:
                    if k == "mfgp_kinv_syrk_f64" and "t64" in name:
                        continue
                    acc.setdefault(k, {}).setdefault(row["Dispatch_Id"], 0.0)
                    acc[k][row["Dispatch_Id"]] += float(row["Counter_Value"])
                    break
    return {k: statistics.median(v.values()) for k, v in acc.items()}


def main():
    n = int(sys.argv[1]); out = sys.argv[2]
    files = dict(a.split("=", 1) for a in sys.argv[3:])
    fetch = per_kernel(files["FETCH_SIZE"], "FETCH_SIZE")
    write = per_kernel(files["WRITE_SIZE"], "WRITE_SIZE")
    Np = (n + 127) // 128 * 128
    alg = {"mfgp_kinv_syrk_f64": 2 * 4 * Np * (Np + 128), "mfgp_kbuild_f64<0>": 4 * Np * (Np + 64) + 8 * n * 5}
    res = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes (with --kernel-trace only) over "
                   "`python3 tools/time_eval.py %d`; values are per launch, median over the launches in the trace. FETCH_SIZE is "
                   "doubled as MI355X_MICROARCH.md (HBM section) prescribes for wide coalesced reads on gfx950; WRITE_SIZE is "
                   "taken as is." % n, "n": n}
    for k in KERNELS:
        if k in fetch or k in write:
            fk, wk = fetch.get(k, 0.0), write.get(k, 0.0)
            res[k] = {"fetch_kb_raw": round(fk), "write_kb": round(wk), "traffic_bytes": int((2 * fk + wk) * 1024)}
            if k in alg:
                res[k]["algorithmic_bytes"] = alg[k]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


main()
