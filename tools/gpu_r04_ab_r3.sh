#!/bin/bash
# A/B on one box: round 3's tree (commit 86548ac) against HEAD, the default bench without the CPU baseline, alternating
out=$GRAFT_REPO_ROOT/gpurun_out/r04ab; mkdir -p $out
for i in 1 2 3; do
  (cd $GRAFT_REPO_ROOT/_r3tree && timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $out/r3_$i.json) || exit 1
  (cd $GRAFT_REPO_ROOT && timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $out/head_$i.json) || exit 1
  python - $out/r3_$i.json $out/head_$i.json <<'PY'
import json, sys
a, b = (json.load(open(f)) for f in sys.argv[1:3])
print("round-3 tree %.1f ms   HEAD %.1f ms" % (a["value"], b["value"]), flush=True)
PY
done
