#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab}; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for n in 2048 4096; do
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$n -- python3 $GRAFT_REPO_ROOT/tools/time_eval.py $n > $out/trace_$n.log 2>&1
(cd $GRAFT_REPO_ROOT && python tools/trace_timeline.py $out/trace_$n 0 400 > $out/timeline_$n.txt 2>&1; python tools/chain_account.py $out/trace_$n > $out/chain_account_$n.txt 2>&1)
cat $out/chain_account_$n.txt
done
