#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-q}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_plans.py tests/test_gpu_multirank.py -m gpu -x -q > $out/tests.log 2>&1; tail -4 $out/tests.log
echo "== bench 2-rank rehearsal on one GPU (TCP collectives)"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 1 --single-device --points 2048 --no-cpu-baseline > $out/bench_n2.json 2> $out/bench_n2.err; python -c "
import json; d=json.loads(open('$out/bench_n2.json').read().strip().splitlines()[-1]); print(d['value'], d['n_gpus'], d['config']['collectives'], d.get('rowblock_allgather'))"; tail -2 $out/bench_n2.err
