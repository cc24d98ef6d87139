"""Summarise rocprofv3 --pmc counter_collection CSVs (one pass per counter) into HBM traffic per launch and per evaluation.
usage: pmc_summary.py N out.json FETCH_SIZE=<csv> WRITE_SIZE=<csv> [ADAPT_FETCH_SIZE=<csv> ADAPT_WRITE_SIZE=<csv> [ADAPT_STATS=<kernel_stats csv>]]

ADAPT_*: the same two passes over `python3 tools/predv_once.py 8192` -- the adaptation loop's own kernels (round 6): the
triangular (multi-)vector products of a predict with N* <= 16 test rows and of a rank-1 append, one entry per kernel
INSTANTIATION (the template arguments are the test rows / rows per wave / chunks per batch), with the algorithmic bytes of one
launch = one read of the 4 Np (Np + 1)-byte lower (or upper) part of the mirrored inverse + the right-hand sides.

FETCH_SIZE / WRITE_SIZE are in KB.  FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes
for wide coalesced reads on gfx950; WRITE_SIZE is taken as is.  Per kernel: the median over its launches.  "sweep": the sum
over every leaf / tile-GEMM dispatch of ONE evaluation (the dispatches between the last two K-build launches of the trace:
tools/time_eval.py runs the same evaluation several times), i.e. the traffic of the Cholesky + inverse + K^-1 stage whose
flops bench.py's `roofline` reports."""
import csv, json, os, sys, statistics

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multifidelity_datafusion_gps_amd.build import source_hash  # noqa: E402  (the sources the profiled library was built from)

KERNELS = ["mfgp_kbuild_rbf2_batch_f64", "mfgp_kbuild_rbf2_f64<0>", "mfgp_kbuild_f64<0>", "mfgp_predvar_f64", "mfgp_grad_rbf2_f64", "mfgp_grad_tiles_f64",
           "mfgp_predv_skinny_f64", "mfgp_kinv_syrk_f64", "mfgp_gemm_nt_f64_t128", "mfgp_gemm_nt_f64_t64",
           "mfgp_gemm_nt_f64_chain", "mfgp_leaf_cholinv_f64", "mfgp_trimv_f64", "mfgp_alpha_finish_f64"]
SWEEP = ("mfgp_gemm_nt_f64_t128", "mfgp_gemm_nt_f64_t64", "mfgp_gemm_nt_f64_chain", "mfgp_leaf_cholinv_f64")


def which(name):
    for k in KERNELS:
        if k in name:
            if k == "mfgp_predvar_f64" and "t64" in name:
                continue
            return k
    return None


def load(path, counter):
    rows = []   # (dispatch id, kernel key, value) with the counter summed over its per-XCD rows
    acc, order = {}, []
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            d = int(row["Dispatch_Id"])
            if d not in acc:
                acc[d] = [which(row["Kernel_Name"]), 0.0]
                order.append(d)
            acc[d][1] += float(row["Counter_Value"])
    for d in sorted(order):
        rows.append((d, acc[d][0], acc[d][1]))
    return rows


def per_kernel(rows):
    by = {}
    for _, k, v in rows:
        if k:
            by.setdefault(k, []).append(v)
    return {k: statistics.median(v) for k, v in by.items()}


def sweep_sum(rows):
    kb = [i for i, (_, k, _) in enumerate(rows) if k and k.startswith("mfgp_kbuild")]
    if len(kb) < 2:
        return None
    return sum(v for _, k, v in rows[kb[-2]:kb[-1]] if k in SWEEP)


ADAPT = ("mfgp_trimv_f64", "mfgp_predv_rows_f64", "mfgp_predv_rows_lds_f64", "mfgp_predv_mfma_f64", "mfgp_predv_mfma2_f64",
         "mfgp_panel_fragments_f64", "mfgp_predv_finish_f64", "mfgp_predv_finish_planes_f64", "mfgp_kpanel_few_rbf2_f64", "mfgp_append_finish_f64", "mfgp_kbuild_rbf2_f64<1>")


def short(name):
    n = name.replace("void ", "").replace("mfgp::", "")
    n = n.split("(")[0].strip()
    return n if any(n.startswith(a) for a in ADAPT) else None


def load_adapt(path, counter):
    acc = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = short(row["Kernel_Name"])
            if k:
                acc.setdefault(k, {}).setdefault(int(row["Dispatch_Id"]), 0.0)
                acc[k][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
    return {k: statistics.median(v.values()) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def adapt_block(n, files):
    Np = (n + 127) // 128 * 128
    tri = 4 * Np * (Np + 1)
    fetch, cnt = load_adapt(files["ADAPT_FETCH_SIZE"], "FETCH_SIZE")
    write, _ = load_adapt(files["ADAPT_WRITE_SIZE"], "WRITE_SIZE")
    dur = {}
    if "ADAPT_STATS" in files and os.path.exists(files["ADAPT_STATS"]):
        with open(files["ADAPT_STATS"]) as f:
            for row in csv.DictReader(f):
                k = short(row["Name"])
                if k:
                    dur[k] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3, "min_us": float(row["MinNs"]) / 1e3}
    out = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes, --kernel-trace only) and rocprofv3 --kernel-trace --stats (a "
                   "third run, no counters: the durations) over `python3 tools/predv_once.py %d` (N = %d - 64 rows, Np = %d): 5 predict calls "
                   "each at N* = 1, 2, 4, 8, 16, 32, 48, 64, then 5 rank-1 appends.  Per kernel instantiation: median over its launches; FETCH_SIZE "
                   "doubled as the guide prescribes for wide coalesced reads.  algorithmic_bytes: one read of the triangle of the mirrored "
                   "inverse (4 Np (Np + 1)) + R right-hand-side rows of 8 Np bytes; achieved = algorithmic bytes / average duration of the "
                   "un-countered run." % (n, n, Np)}
    for k in sorted(set(fetch) | set(write)):
        fk, wk = fetch.get(k, 0.0), write.get(k, 0.0)
        e = {"launches_in_pass": cnt.get(k, 0), "fetch_kb_raw": round(fk), "write_kb": round(wk), "traffic_bytes": int((2 * fk + wk) * 1024)}
        R = None
        if k.startswith(("mfgp_trimv_f64", "mfgp_predv_rows_f64", "mfgp_predv_rows_lds_f64")):
            R = int(k.split("<")[1].split(",")[0])
        elif k.startswith(("mfgp_predv_mfma_f64", "mfgp_predv_mfma2_f64")):       # (the partial planes of the second are not algorithmic)
            R = 16 * int(k.split("<")[1].split(",")[0])
        if R is not None:
            e["algorithmic_bytes"] = tri + R * 8 * Np
        if k in dur:
            e.update(dur[k])
            if "algorithmic_bytes" in e:
                e["achieved_GBps"] = round(e["algorithmic_bytes"] / (dur[k]["avg_us"] * 1e-6) / 1e9, 1)
                e["frac_of_8TBps"] = round(e["achieved_GBps"] / 8000.0, 4)
        out[k] = e
    return out


def main():
    n = int(sys.argv[1]); out = sys.argv[2]
    files = dict(a.split("=", 1) for a in sys.argv[3:])
    fr, wr = load(files["FETCH_SIZE"], "FETCH_SIZE"), load(files["WRITE_SIZE"], "WRITE_SIZE")
    fetch, write = per_kernel(fr), per_kernel(wr)
    Np = (n + 127) // 128 * 128
    alg = {"mfgp_kbuild_rbf2_f64<0>": 4 * Np * (Np + 64) + 8 * n * 5, "mfgp_kbuild_f64<0>": 4 * Np * (Np + 64) + 8 * n * 5,
           "mfgp_predvar_f64": 8 * Np * Np + 4 * Np * Np + 8 * Np * Np}
    res = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes (with --kernel-trace only) over "
                   "`python3 tools/time_eval.py %d`; per kernel: per launch, median over the launches in the trace; 'sweep': sum over "
                   "the leaf + tile-GEMM dispatches of one evaluation. FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) "
                   "prescribes for wide coalesced reads on gfx950; WRITE_SIZE is taken as is." % n, "n": n,
           "csrc_hash": source_hash()}
    for k in KERNELS:
        if k in fetch or k in write:
            fk, wk = fetch.get(k, 0.0), write.get(k, 0.0)
            res[k] = {"fetch_kb_raw": round(fk), "write_kb": round(wk), "traffic_bytes": int((2 * fk + wk) * 1024)}
            if k in alg:
                res[k]["algorithmic_bytes"] = alg[k]
    sf, sw = sweep_sum(fr), sweep_sum(wr)
    if sf is not None and sw is not None:
        res["sweep"] = {"fetch_kb_raw": round(sf), "write_kb": round(sw), "traffic_bytes": int((2 * sf + sw) * 1024),
                        "algorithmic_bytes": 3 * 4 * Np * (Np + 128),
                        "note": "algorithmic = one read of the lower triangle of Ky and one write each of L, L^-1 (mirrored: full) ~ "
                                "K^-1: 12 Np^2 B; the rest is operand panels re-read through L2 / Infinity Cache by the tile GEMMs"}
    if "ADAPT_FETCH_SIZE" in files and "ADAPT_WRITE_SIZE" in files:
        res["adapt"] = adapt_block(n, files)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


main()
