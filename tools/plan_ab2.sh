#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab}; mkdir -p $out
export TMPDIR=/tmp
run() { printf "%-44s" "$1"; env $2 timeout -k 10 120 python tools/time_eval.py $3 2>&1 | awk '{printf "  %s %s", $1, $3}' ; echo; }
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_plans.py -m gpu -x -q > $out/tests.log 2>&1; tail -4 $out/tests.log
{
run "default (fused chain steps, N < 6144)" "A=1" "256 512 1024 2048 3072 4096 6144"
run "FUSE_CHAIN=0" "MFGP_FUSE_CHAIN=0" "256 512 1024 2048 3072 4096"
run "FUSE_CHAIN=1 (all sizes)" "MFGP_FUSE_CHAIN=1" "6144 8192"
} | tee $out/plan_ab2.txt
cd /tmp; timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace_4096 -- python3 $GRAFT_REPO_ROOT/tools/time_eval.py 4096 > $out/trace_4096.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/chain_account.py $out/trace_4096; python tools/trace_timeline.py $out/trace_4096 0 40 | head -40
find $out/trace_4096 -name "*.csv" -size +20M -delete
for c in 2; do
echo "== bench concurrency $c at 4096"
timeout -k 10 300 python bench.py --points 4096 --concurrency $c --no-cpu-baseline > $out/bench4096.json 2> $out/bench.err; python -c "
import json,sys; d=json.loads(open('$out/bench4096.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['wall_ms_per_evaluation'])"; tail -2 $out/bench.err
MFGP_FUSE_CHAIN=0 timeout -k 10 300 python bench.py --points 4096 --concurrency $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nofuse', d['value'], d['config']['wall_ms_per_evaluation'])"
done
