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


# ---- round 3: nothing executable on the wire, admission by job token, a hung RCCL initialisation is fatal -------------
def test_wire_format_round_trips_plain_data_and_refuses_everything_else():
    import pickle
    from multifidelity_datafusion_gps_amd.sharding import wire_decode, wire_encode
    obj = [(1.5, np.arange(6.0).reshape(2, 3), 4), {"a": None, "b": [True, False, "x", b"\x00\x01"]}, (), -7,
           np.array([1, 2, 3], dtype=np.int64), np.zeros((0, 4))]
    back = wire_decode(wire_encode(obj))
    assert back[0][0] == 1.5 and back[0][2] == 4 and isinstance(back[0], tuple)
    np.testing.assert_array_equal(back[0][1], obj[0][1])
    assert back[1] == obj[1] and back[2] == () and back[3] == -7
    np.testing.assert_array_equal(back[4], obj[4])
    assert back[4].dtype == np.int64 and back[5].shape == (0, 4)
    with pytest.raises(TypeError):
        wire_encode({"f": len})                      # a callable does not travel
    with pytest.raises(TypeError):
        wire_encode(np.array([object()]))            # nor does an object array
    with pytest.raises(ValueError):
        wire_decode(pickle.dumps({"r": 1}))          # a pickle is not a frame: rejected, never executed
    with pytest.raises(ValueError):
        wire_decode(wire_encode([1, 2]) + b"x")      # trailing bytes
    with pytest.raises(ValueError):
        wire_decode(b"l" + (2 ** 40).to_bytes(8, "little"))   # a length that the frame cannot hold


def _token_worker(rank, world, port, q, token):
    from multifidelity_datafusion_gps_amd.sharding import SocketComm
    comm = SocketComm(rank, world, "127.0.0.1", port, timeout=30, token=token)
    try:
        res = comm.allgather_object(rank)
        mode = None
        if rank == 0 and getattr(comm, "_token_file", None):
            import os
            import stat
            mode = stat.S_IMODE(os.stat(comm._token_file).st_mode)      # the token file while the job is alive
        modes = comm.allgather_object(mode)
        q.put((rank, res if token is not None else res + [modes[0]]))
    finally:
        comm.close()


def test_rendezvous_ignores_connections_without_the_job_token():
    """a stranger that connects to the hub's port -- with garbage, with a pickle, or with a wrong token -- is dropped
    before anything it sent is decoded, and the job's own ranks still complete the rendezvous"""
    import pickle
    import threading
    import time
    from multifidelity_datafusion_gps_amd.sharding import SocketComm
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    hub = ctx.Process(target=_token_worker, args=(0, 2, port, q, "job-secret"))
    hub.start()
    outcome = {}

    def stranger(kind):
        deadline = time.time() + 20
        while time.time() < deadline:
            try:
                s = socket.create_connection(("127.0.0.1", port), timeout=2.0)
                break
            except OSError:
                time.sleep(0.05)
        else:
            outcome[kind] = "no hub"
            return
        try:
            if kind == "pickle":
                blob = pickle.dumps({"rank": 1})
                s.sendall(len(blob).to_bytes(8, "little") + blob)
                s.settimeout(5.0)
                s.recv(64)            # the hub's challenge; it answers nothing else
                outcome[kind] = "closed" if s.recv(64) == b"" else "answered"
            else:
                SocketComm._introduce(s, b"k" * 32, 1)      # a wrong token
                outcome[kind] = "admitted"
        except (OSError, ValueError, ConnectionError):
            outcome[kind] = "closed"
        finally:
            s.close()

    ts = [threading.Thread(target=stranger, args=(k,)) for k in ("pickle", "wrong-token")]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=30)
    assert outcome == {"pickle": "closed", "wrong-token": "closed"}
    spoke = ctx.Process(target=_token_worker, args=(1, 2, port, q, "job-secret"))
    spoke.start()
    out = dict(q.get(timeout=60) for _ in range(2))
    for p in (hub, spoke):
        p.join(timeout=30)
        assert p.exitcode == 0
    assert out == {0: [0, 1], 1: [0, 1]}


def test_token_file_rendezvous_without_an_explicit_token(tmp_path, monkeypatch):
    """no MFGP_COMM_TOKEN (a plain torch.distributed.run launch): rank 0 writes a 0600 token under a 0700 directory"""
    import stat
    monkeypatch.setenv("XDG_RUNTIME_DIR", str(tmp_path))
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_token_worker, args=(r, 2, port, q, None)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert out == {0: [0, 1, 0o600], 1: [0, 1, 0o600]}                  # (third entry: the token file's mode while the job ran)
    d = tmp_path / "mfgp-comm"
    assert stat.S_IMODE(d.stat().st_mode) == 0o700
    assert not (d / ("token-%d" % port)).exists()                       # rank 0 removed it when it closed


class _HangingEngine(_FakeEngine):
    def __init__(self, hang, rank):
        super().__init__(None, rank)
        self.hang, self.poisoned = hang, None

    def comm_init(self, uid, rank, size):
        if self.hang:
            import time
            time.sleep(3600)
        self.comm_size = size

    def comm_destroy(self):
        self.comm_size = 1

    def poison(self, why):
        self.poisoned = why


def _hang_worker(rank, world, port, q):
    from multifidelity_datafusion_gps_amd.sharding import RcclInitError, SocketComm
    comm = SocketComm(rank, world, "127.0.0.1", port, timeout=60, token="t")
    eng = _HangingEngine(rank == 1, rank)
    try:
        comm.attach_engine(eng, init_timeout=2.0)
        q.put((rank, "attached"))
    except RcclInitError as ex:
        q.put((rank, ("hung" if ex.hung else "peer hung", eng.poisoned is not None, comm.transport)))
    finally:
        comm.close()
    import os
    q.close()
    q.join_thread()  # the queue's feeder thread has flushed
    os._exit(0)      # what a caller has to do where the initialisation is stuck


def test_hung_rccl_initialisation_raises_on_every_rank_and_poisons_the_stuck_handle():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_hang_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    assert out[1] == ("hung", True, "tcp")           # the rank whose call never returned: handle abandoned
    assert out[0] == ("peer hung", False, "tcp")     # the other rank raises too: nobody continues half-attached


def test_required_attach_raises_instead_of_falling_back():
    from multifidelity_datafusion_gps_amd.sharding import RcclInitError, SocketComm
    comm = SocketComm(0, 1)
    comm.size = 1            # a one-rank communicator still goes through the agreement
    with pytest.raises(RcclInitError):
        class _E(_FakeEngine):
            def comm_init(self, uid, rank, size):
                raise RuntimeError("ncclCommInitRank: invalid usage (test)")
        comm.attach_engine(_E(), required=True)


# ---- the sequential pieces run on rank 0 only and are adopted by the others -----------------------------------------------
def _run_datalf_model(comm, restarts):
    import multifidelity_datafusion_gps_amd as mf
    from tests.oracle_engine import OracleEngine
    from tests.test_sharding_gloo import hf, lf
    rng = np.random.default_rng(3)
    X_lf = rng.uniform(size=(40, 2))
    engines = {k: OracleEngine() for k in ("lf", "hf", "hf#1", "hf#2")}

    class M(mf.NARGP):
        lf_max_iters = first_run_max_iters = restart_max_iters = 20
        num_restarts = restarts
        restart_concurrency = 2

    model = M(2, hf, None, lf_X=X_lf, lf_Y=lf(X_lf), seed=5, comm=comm, engines=engines, device_chaining=False)
    model.fit(rng.uniform(size=(20, 2)))
    mean, var = model.predict(rng.uniform(size=(37, 2)))
    return dict(lf_theta=np.array([p.value for p in model.lf_model.parameters()]),
                theta=np.array([p.value for p in model.hf_model.parameters()]), mean=mean, var=var,
                lf_evals=engines["lf"].n_evals)


def _datalf_worker(rank, world, port, q, restarts):
    from multifidelity_datafusion_gps_amd.sharding import SocketComm
    comm = SocketComm(rank, world, "127.0.0.1", port, timeout=60, token="t")
    try:
        q.put((rank, _run_datalf_model(comm, restarts)))
    finally:
        comm.close()


@pytest.mark.parametrize("restarts", [6, 0])
def test_low_fidelity_run_happens_on_rank0_only_and_every_rank_ends_in_the_same_state(restarts):
    """data-driven low-fidelity level on 3 ranks: only rank 0 runs its L-BFGS-B, the others adopt the optimum and factorise
    once; with no restarts at all (restarts = 0) the ranks that never ran the first high-fidelity run adopt rank 0's optimum
    too: every rank's predictions are the single-process ones."""
    from multifidelity_datafusion_gps_amd.sharding import LocalComm
    ref = _run_datalf_model(LocalComm(), restarts)
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_datalf_worker, args=(r, 3, port, q, restarts)) for r in range(3)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0]["lf_evals"] > 5 and out[1]["lf_evals"] <= 2 and out[2]["lf_evals"] <= 2     # one factorisation, no run
    for r in range(3):
        np.testing.assert_array_equal(out[r]["lf_theta"], out[0]["lf_theta"])
        np.testing.assert_array_equal(out[r]["theta"], out[0]["theta"])
        np.testing.assert_array_equal(out[r]["mean"], out[0]["mean"])
        np.testing.assert_allclose(out[r]["theta"], ref["theta"], rtol=1e-10)
        np.testing.assert_allclose(out[r]["mean"], ref["mean"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(out[r]["var"], ref["var"], rtol=0, atol=1e-6)


def test_restart_assignment_on_eight_ranks():
    """the bench's HF level on 8 ranks: rank 0 keeps the sequential first run -> restart 0, the five randomized restarts go
    one each to the highest ranks, ranks 1-2 have no optimiser run (they adopt the winner and shard the predict)"""
    from multifidelity_datafusion_gps_amd.abstractMFGP import AbstractMFGP
    a = AbstractMFGP.assign_restarts(6, 8)
    assert a == [[], [], [], [5], [4], [3], [2], [1]]
    assert AbstractMFGP.assign_restarts(6, 2) == [[4], [1, 2, 3, 5]] and AbstractMFGP.assign_restarts(6, 1) == [[1, 2, 3, 4, 5]]
    assert AbstractMFGP.assign_restarts(6, 4) == [[], [3], [2, 5], [1, 4]]


def test_wire_format_edge_cases():
    """0-d arrays keep their shape, out-of-range ints are refused on the sending side, every malformed frame is a
    ValueError on the receiving side (truncation, absurd nesting, absurd array headers)"""
    import struct
    from multifidelity_datafusion_gps_amd.sharding import wire_decode, wire_encode
    a0 = np.array(3.5)
    back = wire_decode(wire_encode(a0))
    assert back.shape == () and back == 3.5
    e = np.zeros((0, 4))
    assert wire_decode(wire_encode(e)).shape == (0, 4)
    with pytest.raises(TypeError):
        wire_encode(2 ** 63)
    with pytest.raises(TypeError):
        wire_encode(-2 ** 63 - 1)
    assert wire_decode(wire_encode(2 ** 63 - 1)) == 2 ** 63 - 1
    deep = b"l" + struct.pack("<Q", 1)
    with pytest.raises(ValueError):
        wire_decode(deep * 2000 + b"N")
    with pytest.raises(ValueError):
        wire_decode(b"a" + struct.pack("<BB", 0, 2) + struct.pack("<2Q", 0, 2 ** 63))      # (0, 2^63) array header
    with pytest.raises(ValueError):
        wire_decode(b"i" + b"\x00" * 3)                                                     # truncated int
    with pytest.raises(ValueError):
        wire_decode(b"s" + struct.pack("<Q", 2) + b"\xff\xfe")                             # not UTF-8
