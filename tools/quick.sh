#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-q}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_plans.py tests/test_gpu_multirank.py -m gpu -x -q > $out/tests.log 2>&1; tail -4 $out/tests.log
for n in 2 4; do
echo "== bench $n-rank rehearsal on one GPU (TCP collectives)"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2953$n bench.py --gpus $n --steps 1 --warmup 1 --single-device --points 2048 --no-cpu-baseline > $out/bench_n$n.json 2> $out/bench_n$n.err; python -c "
import json; d=json.loads(open('$out/bench_n$n.json').read().strip().splitlines()[-1]); print(d['value'], d['n_gpus'], d['config']['collectives'], d.get('rowblock_allgather'), d['result_checksum'])"; tail -2 $out/bench_n$n.err
done
timeout -k 10 300 python bench.py --steps 1 --warmup 1 --points 2048 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['n_gpus'], d['result_checksum'])"
