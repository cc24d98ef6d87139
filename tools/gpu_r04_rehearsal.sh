#!/bin/bash
# round 4: the N-rank bench at FULL size on the one-GPU box (every rank on GPU 0, RCCL over its socket transport -- sharding.rehearsal_env):
# the code path the driver's 2 / 4 / 8-GPU runs take (sharded sequential evaluations, lock-stepped restarts per rank, sharded predict), end to
# end at N = 8192.  The VALUES are meaningless as performance (N ranks share one GPU and a loopback socket carries the exchanges).
set -o pipefail
out=gpurun_out/${1:-r04reh}; mkdir -p $out
for n in 2 4; do
  timeout -k 10 500 python bench.py --gpus $n --single-device --no-cpu-baseline --steps 1 --warmup 1 > $out/bench_n$n.json 2> $out/bench_n$n.err || { tail -20 $out/bench_n$n.err; exit 1; }
  python -c "
import json; d=json.loads([l for l in open('$out/bench_n$n.json') if l.startswith('{')][-1])
print('$n ranks on one GPU:', d['value'], 'ms; rccl_ranks', d['config']['rccl_ranks'], '|', d['config']['collectives'][:100])"
done
# the driver's own launch form (torch.distributed.run starts the ranks; bench.py reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --single-device --no-cpu-baseline --steps 1 --warmup 1 > $out/bench_n2_torchrun.json 2> $out/bench_n2_torchrun.err || { tail -20 $out/bench_n2_torchrun.err; exit 1; }
python -c "
import json; d=json.loads([l for l in open('$out/bench_n2_torchrun.json') if l.startswith('{')][-1])
print('torchrun, 2 ranks on one GPU:', d['value'], 'ms; rccl_ranks', d['config']['rccl_ranks'])"
