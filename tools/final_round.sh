#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-fin}; mkdir -p $out
echo "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== gpu tests"; timeout -k 10 900 python -m pytest tests -m gpu -q > $out/tests.log 2>&1; tail -3 $out/tests.log
bash tools/gpu_profile.sh $1 2>&1 | tail -60
bash tools/gpu_traces.sh $1 2>&1 | tail -30
