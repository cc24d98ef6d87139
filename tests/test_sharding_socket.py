"""CPU tests of the torch-free host communicator of the multi-GPU path (sharding.SocketComm: TCP hub on 127.0.0.1):
its collectives, and the sharded fit + predict of the model through it with the oracle double as the engine
(the same check tests/test_sharding_gloo.py makes through gloo)."""
import multiprocessing as mp
import socket

import numpy as np
import pytest

from tests.test_sharding_gloo import _run_model


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, conc):
    from multifidelity_datafusion_gps_amd.sharding import SocketComm
    comm = SocketComm(rank, world, "127.0.0.1", port, timeout=60)
    try:
        assert comm.allgather_object({"r": rank}) == [{"r": r} for r in range(world)]
        assert comm.bcast_object("x%d" % rank, src=world - 1) == "x%d" % (world - 1)
        comm.barrier()
        g = comm.allgather_rows(np.full((rank + 2, 3), float(rank)))            # ragged blocks
        assert g.shape == (sum(r + 2 for r in range(world)), 3)
        assert [float(v) for v in g[:, 0]] == [float(r) for r in range(world) for _ in range(r + 2)]
        assert comm.transport == "tcp"
        q.put((rank, _run_model(comm, conc) if conc else None))
    finally:
        comm.close()


def _spawn(world, conc):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, conc)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def test_socket_comm_collectives_three_ranks():
    out = _spawn(3, 0)
    assert sorted(out) == [0, 1, 2]


@pytest.mark.parametrize("conc", [1, 2])
def test_two_rank_fit_predict_over_socket_comm_equals_single_process(conc):
    from multifidelity_datafusion_gps_amd.sharding import LocalComm
    ref = _run_model(LocalComm(), conc)
    out = _spawn(2, conc)
    for r in (0, 1):
        np.testing.assert_allclose(out[r]["theta"], ref["theta"], rtol=1e-12)   # same winner on every rank
        np.testing.assert_allclose(out[r]["mean"], ref["mean"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(out[r]["var"], ref["var"], rtol=0, atol=1e-6)
        np.testing.assert_array_equal(out[r]["mean"], out[0]["mean"])
    assert out[0]["evals"] < ref["evals"] and out[1]["evals"] < ref["evals"]


class _FakeEngine:
    """stands in for _lib.Engine in attach_engine: RCCL 'works' (the gather is emulated through a shared list) or fails"""

    def __init__(self, fail_on_rank=None, rank=0):
        self.fail = fail_on_rank == rank
        self.comm_size = 1

    def comm_unique_id(self):
        return bytes(128)

    def comm_init(self, uid, rank, size):
        assert len(uid) == 128
        if self.fail:
            raise RuntimeError("ncclCommInitRank: unhandled system error (test)")
        self.comm_size = size


def _attach_worker(rank, world, port, q, fail_on_rank):
    from multifidelity_datafusion_gps_amd.sharding import SocketComm
    comm = SocketComm(rank, world, "127.0.0.1", port, timeout=60)
    try:
        ok = comm.attach_engine(_FakeEngine(fail_on_rank, rank))
        # a failure on ANY rank keeps EVERY rank on TCP (a half-attached job would hang in its first collective)
        g = comm.allgather_rows(np.full((2, 2), float(rank))) if not ok else None
        q.put((rank, (ok, comm.transport, getattr(comm, "rccl_error", None), None if g is None else g[:, 0].tolist())))
    finally:
        comm.close()


@pytest.mark.parametrize("fail_on_rank", [None, 1])
def test_attach_engine_is_all_or_nothing(fail_on_rank):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_attach_worker, args=(r, 2, port, q, fail_on_rank)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        ok, transport, err, rows = out[r]
        if fail_on_rank is None:
            assert ok and transport == "rccl"
        else:
            assert not ok and transport == "tcp" and "ncclCommInitRank" in err
            assert rows == [0.0, 0.0, 1.0, 1.0]


def test_comm_from_env_single_process_is_local(monkeypatch):
    from multifidelity_datafusion_gps_amd import sharding
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert isinstance(sharding.comm_from_env(), sharding.LocalComm)
