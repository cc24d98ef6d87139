#!/bin/bash
# the default bench line (with the CPU baseline), its wall time, and what the line says
out=gpurun_out/${1:-r04bench}; mkdir -p $out
t0=$(date +%s)
timeout -k 10 800 python bench.py > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
echo "wall $(( $(date +%s) - t0 )) s"
python - "$out/bench.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print("value", d["value"], d["unit"], "| roofline", r["achieved"], r["frac"], "traffic", r["traffic"], "mfma_busy", r["mfma_busy"])
print("sharding:", d["config"]["sharding"])
cb = d["cpu_baseline"]; print("cpu", cb["value"], cb["unit"], cb["cores"], cb["kind"], cb["sample"][:100])
PY
