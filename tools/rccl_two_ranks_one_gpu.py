"""Can RCCL run a communicator of TWO ranks on the ONE GPU of the box?  It refuses two ranks on one device when it sees that
they share a bus id on the same host -- the host identity is NCCL_HOSTID when set, so two ranks that name different hosts are
taken for two machines and talk over the socket transport (loopback).  If that works, the >= 2-rank code path of
csrc/comm_rccl.hip (ncclCommInitRank with a shared unique id, ncclAllGather in place on the device matrix, the host-staged
gather) runs for real on the one-GPU box.  usage: python tools/rccl_two_ranks_one_gpu.py [world=2]"""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, q):
    from multifidelity_datafusion_gps_amd import sharding
    os.environ.update(sharding.rehearsal_env(rank))
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    from multifidelity_datafusion_gps_amd._lib import Engine
    from tests import cases
    comm = sharding.SocketComm(rank, world, "127.0.0.1", port, timeout=120)
    res = {"rank": rank}
    try:
        rng = np.random.default_rng(3)
        X = rng.uniform(size=(700, 4))
        Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
        e = Engine(0)
        e.set_data(Xa, cases.hf_4d(X))
        e.set_kernel(cases.composite(4, 1))
        theta, noise = np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.02
        f0, g0 = e.eval(theta, noise, 1e-8)
        try:
            comm.attach_engine(e, required=True, init_timeout=60)
            res["transport"] = comm.transport
            res["comm_size"] = int(e.comm_size)
            got = e.allgather_host(np.arange(6.0) + 10 * rank)
            res["allgather_host_ok"] = bool(np.array_equal(got, np.arange(6.0)[None, :] + 10 * np.arange(world)[:, None]))
            f1, g1 = sharding.eval_rowblock_allgather(e, comm, theta, noise)
            res["rowblock_bitwise"] = bool(f1 == f0 and np.array_equal(g1, g0))
            comm.barrier()
            e.comm_destroy()
        except sharding.RcclInitError as ex:
            res["error"] = str(ex)[:500]
        comm.barrier()
        e.close()
    finally:
        q.put(res)
        comm.close()


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in sorted(out, key=lambda d: d["rank"]):
        print(r)
    print("exit codes", [p.exitcode for p in procs])


if __name__ == "__main__":
    main()
