"""Summarise the round-3 rocprofv3 --pmc passes (tools/gpu_r03_counters.sh): per kernel, the SQ / GRBM counters summed
over its dispatches next to the dispatches' durations from the kernel trace of the SAME pass.
usage: mfma_counters.py <dir with pmc{A,B,C}_{probes,eval}.csv and *_trace.csv> [out.json]

Derived columns (see profiles/README.md for the calibration on the bare probes):
  clock_ghz   = GRBM_GUI_ACTIVE / 8 XCDs / duration                      (MI355X_MICROARCH.md "DVFS give-back")
  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)  (matrix pipe busy while the CU holds waves)
  mfma_busy_g = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)   (matrix pipe busy over the whole launch)
For the evaluation passes only the dispatches of the LAST evaluation in the trace count (between the last two K builds)."""
import csv, glob, json, os, sys

d = sys.argv[1]


def short(name):
    n = name.replace("void ", "").replace("mfgp::", "")
    if "mfgp_probe_fp64_shape" in n or "mfgp_probe_mfma_detail" in n or "mfgp_kbuild" in n or "mfgp_predv_skinny" in n:
        return n.split("(")[0]
    return n.split("(")[0].split("<")[0]


def load(tag):
    pmc, trace = os.path.join(d, tag + ".csv"), os.path.join(d, tag + "_trace.csv")
    if not os.path.exists(pmc):
        return None
    disp = {}
    with open(pmc) as f:
        for r in csv.DictReader(f):
            k = int(r["Dispatch_Id"])
            e = disp.setdefault(k, {"name": short(r["Kernel_Name"]), "c": {}, "grid": int(r.get("Grid_Size", 0) or 0),
                                    "wg": int(r.get("Workgroup_Size", 0) or 0)})
            e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if "Start_Timestamp" in r and r["Start_Timestamp"]:
                e["t"] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
    if os.path.exists(trace):
        with open(trace) as f:
            for r in csv.DictReader(f):
                k = int(r["Dispatch_Id"])
                if k in disp:
                    disp[k]["t"] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
    return disp


def last_eval(disp):
    ids = sorted(disp)
    kb = [i for i in ids if disp[i]["name"].startswith("mfgp_kbuild")]
    if len(kb) < 2:
        return ids
    return [i for i in ids if kb[-2] <= i < kb[-1]]


def table(disp, ids, per_dispatch=False):
    rows = {}
    for i in ids:
        e = disp[i]
        key = e["name"] + (" #%d" % i if per_dispatch else "")
        r = rows.setdefault(key, {"n": 0, "ns": 0, "c": {}})
        r["n"] += 1
        if "t" in e:
            r["ns"] += e["t"][1] - e["t"][0]
        for c, v in e["c"].items():
            r["c"][c] = r["c"].get(c, 0.0) + v
    out = {}
    for k, r in rows.items():
        c = r["c"]
        o = {"launches": r["n"], "duration_ms": round(r["ns"] / 1e6, 4)}
        o.update({a: b for a, b in c.items()})
        g = c.get("GRBM_GUI_ACTIVE")
        if g and r["ns"]:
            o["clock_ghz"] = round(g / 8.0 / r["ns"], 3)
        mb = c.get("SQ_VALU_MFMA_BUSY_CYCLES")
        if mb is not None and c.get("SQ_BUSY_CU_CYCLES"):
            o["mfma_busy"] = round(mb / (4.0 * c["SQ_BUSY_CU_CYCLES"]), 4)
        if mb is not None and g:
            o["mfma_busy_g"] = round(mb / (1024.0 * g / 8.0), 4)
        out[k] = o
    return out


res = {}
for tag in ("pmcA_probes", "pmcA_eval", "pmcB_eval", "pmcC_eval"):
    disp = load(tag)
    if disp is None:
        continue
    if tag.endswith("probes"):
        res[tag] = table(disp, sorted(disp), per_dispatch=True)
    else:
        res[tag] = table(disp, last_eval(disp))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multifidelity_datafusion_gps_amd.build import source_hash  # noqa: E402
res["csrc_hash"] = source_hash()      # the sources the profiled library was built from (bench.py checks it against mfgp_build_id)
print(json.dumps(res, indent=1))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
