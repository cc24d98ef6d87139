"""Multi-rank tests with the REAL HIP engine: several processes share GPU 0 of the one-GPU box.

* over TCP: RCCL refuses two ranks on one device as it finds them, so the first two-process test moves the device data of
  the row-block layout through host memory (mfgp_rows_download / mfgp_rows_upload) and the small gathers over
  sharding.SocketComm;
* over RCCL, for real: with sharding.rehearsal_env every rank names a host of its own (NCCL_HOSTID), RCCL takes them for
  separate machines and builds a communicator of 2 / 3 ranks over its socket transport -- ncclCommInitRank on the shared
  unique id, the in-place ncclAllGather of the K row blocks between processes, the host-staged gather of the predictive rows;
* the RCCL calls with a communicator of size 1.
Covers e1 (predictive rows sharded), e2 (restarts sharded) and e3 (K row blocks + all-gather + prebuilt evaluation)."""
import multiprocessing as mp
import socket

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu


def hf(x):
    return cases.hf_2d(x)[:, None]


def lf(x):
    return cases.lf_2d(x)[:, None]


def _model_run(comm, conc, restarts=6):
    import multifidelity_datafusion_gps_amd as mf

    class Budget(mf.NARGP):
        lf_max_iters = first_run_max_iters = restart_max_iters = 12
        eval_cap = 12
        restart_concurrency = conc
        num_restarts = restarts

    rng = np.random.default_rng(7)
    X_lf = rng.uniform(size=(300, 2))
    model = Budget(2, hf, None, lf_X=X_lf, lf_Y=lf(X_lf), seed=11, comm=comm)
    lf_theta = np.array([p.value for p in model.lf_model.parameters()])
    lf_evals = model.lf_model.n_evals           # evaluations this rank's low-fidelity model issued for its one optimize() run
    model.fit(rng.uniform(size=(200, 2)))
    mean, var = model.predict(rng.uniform(size=(333, 2)))
    theta = np.array([p.value for p in model.hf_model.parameters()])
    evals = model.hf_model.n_evals
    groups = sorted((g.size, g.index) for _, _, g in getattr(model, "_shard_groups", []) if g is not None)
    # cfg5's panel form: one predictive-variance panel per acquisition, its rows sharded over the ranks and gathered; every
    # rank must acquire the same point
    model.adapt_maximizer = mf.adaptation_maximizers.PanelMaximizer(n_candidates=4096, seed=5)
    model.adapt(2, reoptimize=False)
    acquired = np.array(model.acquired_points).reshape(2, 2)
    model.close()
    return dict(theta=theta, mean=mean, var=var, evals=evals, acquired=acquired, groups=groups,
                lf_theta=lf_theta, lf_evals=lf_evals)


def _rowblock_run(comm):
    from multifidelity_datafusion_gps_amd._lib import Engine
    from multifidelity_datafusion_gps_amd.sharding import eval_rowblock_allgather
    rng = np.random.default_rng(3)
    X = rng.uniform(size=(700, 4))
    Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
    e = Engine(0)
    e.set_data(Xa, cases.hf_4d(X))
    e.set_kernel(cases.composite(4, 1))
    theta, noise = np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.02
    fused = e.eval(theta, noise, 1e-8)
    sharded = eval_rowblock_allgather(e, comm, theta, noise)      # 768 padded rows -> 2 x 384
    e.close()
    return fused, sharded


def _worker(rank, world, port, q):
    from multifidelity_datafusion_gps_amd.sharding import SocketComm
    comm = SocketComm(rank, world, "127.0.0.1", port, timeout=120)
    try:
        q.put((rank, dict(seq=_model_run(comm, 1), conc=_model_run(comm, 2), rowblock=_rowblock_run(comm))))
    finally:
        comm.close()


def test_two_processes_on_one_gpu_with_the_hip_engine():
    from multifidelity_datafusion_gps_amd.sharding import LocalComm
    ref = _model_run(LocalComm(), 1)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        for key in ("seq", "conc"):
            got = out[r][key]
            # e2: the same restarts ran (seeded per restart), split over the ranks; the same winner everywhere
            np.testing.assert_allclose(got["theta"], ref["theta"], rtol=1e-9)
            # e1: row-sharded predictions, gathered: every rank holds the full result, equal to the single process
            np.testing.assert_allclose(got["mean"], ref["mean"], rtol=0, atol=1e-7)
            np.testing.assert_allclose(got["var"], ref["var"], rtol=0, atol=1e-7)
            np.testing.assert_array_equal(got["mean"], out[0][key]["mean"])
            np.testing.assert_array_equal(got["acquired"], ref["acquired"])     # sharded panels: the single process's picks
        assert out[r]["seq"]["evals"] < ref["evals"]
        # e3: K built by two ranks' row blocks + gathered = the fused evaluation, bit for bit (same kernels, same order)
        (f0, g0), (f1, g1) = out[r]["rowblock"]
        assert f1 == f0 and np.array_equal(g1, g0)


def _rccl_worker(rank, world, port, q, dist_everywhere=False):
    import os
    from multifidelity_datafusion_gps_amd import sharding
    os.environ.update(sharding.rehearsal_env(rank))       # before librccl is loaded (lazily, by attach_engine)
    if dist_everywhere:
        os.environ["MFGP_DIST_CHOL"] = "1"                 # every shared evaluation of this worker with the Cholesky distributed too
    from multifidelity_datafusion_gps_amd._lib import Engine
    comm = sharding.SocketComm(rank, world, "127.0.0.1", port, timeout=120)
    res = {}
    try:
        e = Engine(0)
        comm.attach_engine(e, required=True, init_timeout=90)     # RcclInitError on every rank if the communicator fails
        res["transport"], res["comm_size"] = comm.transport, int(e.comm_size)
        got = e.allgather_host(np.arange(7.0) + 100.0 * rank)
        res["host_gather"] = got
        # ragged row counts through the communicator's own row gather (pads to the longest block, trims after)
        res["ragged"] = comm.allgather_rows(np.full((3 + 2 * rank, 2), float(rank)))
        rng = np.random.default_rng(3)
        X = rng.uniform(size=(700, 4))
        Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
        e.set_data(Xa, cases.hf_4d(X))
        e.set_kernel(cases.composite(4, 1))
        theta, noise = np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.02
        res["fused"] = e.eval(theta, noise, 1e-8)
        res["sharded"] = sharding.eval_rowblock_allgather(e, comm, theta, noise)   # 768 padded rows: 6 blocks of 128 over the ranks
        res["model"] = _model_run(comm, 2)                 # e1 + e2 with the row gathers on RCCL
        # two restarts only: with three ranks one of them is dealt none and SHARES first run -> restart 0 with rank 0 (chain group)
        res["model_r2"] = _model_run(comm, 2, restarts=2)
        if world == 3:
            # a job that builds a fresh model per fit on the SAME engine handles (bench.py does, per timed step) forms its groups once:
            # the second fit reuses the chain group's communicator instead of paying ncclCommInitRank inside the fit
            import multifidelity_datafusion_gps_amd as mf
            shared = {k: Engine(0) for k in ("lf", "hf")}
            formed0 = getattr(comm, "groups_formed", 0)
            fits = []
            for _ in range(2):
                class B2(mf.NARGP):
                    lf_max_iters = first_run_max_iters = restart_max_iters = 8
                    eval_cap = 8
                    num_restarts = 2
                rng2 = np.random.default_rng(17)
                X_lf2 = rng2.uniform(size=(300, 2))
                mdl = B2(2, hf, None, lf_X=X_lf2, lf_Y=lf(X_lf2), seed=3, comm=comm, engines=shared)
                mdl.fit(rng2.uniform(size=(200, 2)))
                fits.append(np.array([p.value for p in mdl.hf_model.parameters()]))
            res["group_reuse"] = dict(formed=getattr(comm, "groups_formed", 0) - formed0, same_fit=bool(np.array_equal(fits[0], fits[1])))
            for e_ in shared.values():
                if e_.comm_size > 1:
                    e_.comm_destroy()
            comm.barrier()
            for e_ in shared.values():
                e_.close()
        # ONE evaluation sharded over the ranks (mfgp_eval_sharded: rows of L^-T and of K^-1 by 128-row block, grouped
        # ncclBroadcast of the rows + ncclAllReduce of the gradient's tile sums): bitwise the single evaluation, at a size with
        # several macro panels and at the north-star size
        res["eval_sharded"] = {}
        for n in ((2100, 8192) if world <= 3 else (2100,)):
            rng = np.random.default_rng(n)
            X = rng.uniform(size=(n, 4))
            Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
            Y = cases.hf_4d(X)
            e.set_data(Xa, Y)
            e.set_kernel(cases.composite(4, 1))
            th, nz = np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.01 * Y.var()
            f0, g0 = e.eval(th, nz, 1e-8)
            m0, v0 = e.predict(Xa[:100])
            comm.barrier()
            f1, g1 = e.eval_sharded(th, nz, 1e-8)
            m1, v1 = e.predict(Xa[:100])                  # the exchanged factor serves predictions on every rank
            f2 = e.eval_sharded(th * 1.05, nz, 1e-8, want_grad=False)
            f3 = e.eval(th * 1.05, nz, 1e-8, want_grad=False)
            res["eval_sharded"][n] = dict(single=(f0, g0, m0, v0), sharded=(f1, g1, m1, v1), nograd=(f2, f3))
        # a kernel structure outside the RBF fast path (Matern factors, ARD): the generic gradient kernel's ownership filter
        e.set_kernel(cases.composite(4, 1, cases.M52, cases.RBF | cases.ARD, cases.M32))
        thg = np.array([1.1, 1.3, 0.7, 0.8, 0.6, 1.2, 0.9, 0.5, 0.9])
        f0, g0 = e.eval(thg, nz, 1e-8)
        m0, v0 = e.predict(Xa[:100])
        comm.barrier()
        f1, g1 = e.eval_sharded(thg, nz, 1e-8)
        m1, v1 = e.predict(Xa[:100])
        res["eval_sharded"]["matern_ard_%d" % len(Y)] = dict(single=(f0, g0, m0, v0), sharded=(f1, g1, m1, v1), nograd=(0.0, 0.0))
        # SURVEY 8(e) "Cholesky": the factorisation itself distributed over the group (1-D block-cyclic rows, plan.h Shard::dist; the size
        # rule turns it on from N = 16384, MFGP_DIST_CHOL=1 here): every rank factorises only the diagonal blocks it owns and runs only
        # its rows of the panels and trailing updates; the diagonal blocks and panel columns travel (ncclBroadcast / ncclAllGather per
        # block column).  Bitwise the single evaluation -- NLML, gradient, the factor itself, predictions -- and a matrix that is not
        # positive definite reports the same pivot on every rank.
        if world <= 3:
            os.environ["MFGP_DIST_CHOL"] = "1"             # (read when the handle plans: at the next set_data)
            res["dist"] = {}
            for n in (900, 4096):
                rng = np.random.default_rng(n)
                X = rng.uniform(size=(n, 4))
                Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
                Y = cases.hf_4d(X)
                e.set_data(Xa, Y)
                e.set_kernel(cases.composite(4, 1))
                th, nz = np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.01 * Y.var()
                f0, g0 = e.eval(th, nz, 1e-8)
                L0 = e.get_L()
                m0, v0 = e.predict(Xa[:100])
                comm.barrier()
                f1, g1 = e.eval_sharded(th, nz, 1e-8)
                L1 = e.get_L()
                m1, v1 = e.predict(Xa[:100])
                res["dist"][n] = dict(single=(f0, g0, m0, v0), sharded=(f1, g1, m1, v1), L_equal=bool(np.array_equal(L0, L1)))
                if n == 900:       # the leader / follower form runs the same distributed plan (one optimiser, the others serve)
                    comm.barrier()
                    if rank == 0:
                        f2, g2 = e.sharded_lead(th * 1.03, nz, 1e-8)
                        e.sharded_release()
                    else:
                        assert e.sharded_serve() == 1
                        f2, g2 = None, None
                    f3, g3 = e.eval(th * 1.03, nz, 1e-8)
                    res["dist"]["lead"] = (f2, g2, f3, g3)
            if world == 2 and not dist_everywhere:
                # ... and WITHOUT a switch in the environment the planner decides from what a collective of this group was MEASURED to
                # cost when the communicator was formed (comm.attach_engine -> Engine.comm_calibrate): on this rig --
                # two ranks on one GPU over RCCL's socket transport -- a collective costs far more than the 2 nb - 1 of them save,
                # so the north-star size is planned with the Cholesky REPLICATED, and the decision says why
                del os.environ["MFGP_DIST_CHOL"]
                rng = np.random.default_rng(8192)
                Xb = rng.uniform(size=(8192, 4))
                Xba = np.hstack([Xb, cases.lf_4d(Xb)[:, None]])
                Yb = cases.hf_4d(Xb)
                e.set_data(Xba, Yb)
                e.set_kernel(cases.composite(4, 1))
                nzb = 0.01 * Yb.var()
                f0, g0 = e.eval(th, nzb, 1e-8)
                comm.barrier()
                f1, g1 = e.eval_sharded(th, nzb, 1e-8)
                res["dist"]["measured_8192"] = (f0, g0, f1, g1)
                res["dist"]["calibration"] = comm.calibration
                res["dist"]["decision_8192"] = e.shard_decision()
                # ... the distributed plan at N = 16384 (128 block columns, 255 exchange steps), switched on; skipped where the device
                # has no room for two ranks' slabs
                os.environ["MFGP_DIST_CHOL"] = "1"
                free, _ = e.mem_info()
                if free > 2 * (4 * 16384 ** 2 * 8) + (8 << 30):
                    rng = np.random.default_rng(16384)
                    Xb = rng.uniform(size=(16384, 4))
                    Xba = np.hstack([Xb, cases.lf_4d(Xb)[:, None]])
                    Yb = cases.hf_4d(Xb)
                    e.set_data(Xba, Yb)
                    e.set_kernel(cases.composite(4, 1))
                    nzb = 0.01 * Yb.var()
                    f0, g0 = e.eval(th, nzb, 1e-8)
                    comm.barrier()
                    f1, g1 = e.eval_sharded(th, nzb, 1e-8)
                    res["dist"]["forced_16384"] = (f0, g0, f1, g1)
                    res["dist"]["decision_16384"] = e.shard_decision()
                else:
                    res["dist"]["forced_16384"] = "skipped: %.1f GB free" % (free / 1e9)
            Xd = np.vstack([Xa[:700], Xa[300:600]])        # duplicated rows, no noise, no jitter: not positive definite
            e.set_data(Xd, np.concatenate([Y[:700], Y[300:600]]))
            e.set_kernel(cases.composite(4, 1))
            pivots = []
            for fn in (e.eval, e.eval_sharded):
                comm.barrier()
                try:
                    fn(th, 0.0, 0.0)
                    pivots.append(0)
                except Exception as ex:  # noqa: BLE001 - NotPositiveDefinite carries the pivot
                    pivots.append(int(getattr(ex, "info", -1)))
            res["dist"]["not_pd"] = pivots
            if not dist_everywhere:
                del os.environ["MFGP_DIST_CHOL"]
        comm.barrier()
        e.comm_destroy()                                   # every rank still alive
        comm.barrier()
        e.close()
    finally:
        q.put((rank, res))
        comm.close()


@pytest.mark.parametrize("world,dist_everywhere", [(2, False), (3, False), (5, False), (2, True)],
                         ids=["2", "3", "5", "2-distributed-cholesky-everywhere"])
def test_rccl_communicator_of_several_ranks_on_one_gpu(world, dist_everywhere):
    """A REAL RCCL communicator with more than one rank ("an RCCL collective with >= 2 ranks has never executed
    anywhere"): mfgp_comm_unique_id on rank 0 -> TCP -> mfgp_comm_init on every rank, then mfgp_allgather_rows (ONE
    in-place ncclAllGather on the device matrix) and mfgp_allgather_host between the processes.  world = 5 is the most a one-GPU box
    admits beside this test's own process, which computes the single-process reference on the same card (its process guard stops at 6;
    `bench.py --gpus 6 --single-device`, whose launcher never touches the GPU, is the 6-rank rehearsal: profiles/r05_bench_n6_*);
    the 8-rank layout runs on the CPU (tests/test_bench_launcher.py) and in the driver's job.  The last case runs everything with
    MFGP_DIST_CHOL=1: every shared evaluation -- the models' low-fidelity runs, the chain group's, N = 8192 -- over the distributed plan."""
    from multifidelity_datafusion_gps_amd.sharding import LocalComm
    ref = _model_run(LocalComm(), 1)
    ref_r2 = _model_run(LocalComm(), 1, restarts=2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, q, dist_everywhere)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        o = out[r]
        assert o["transport"] == "rccl" and o["comm_size"] == world
        np.testing.assert_array_equal(o["host_gather"], np.arange(7.0)[None, :] + 100.0 * np.arange(world)[:, None])
        np.testing.assert_array_equal(o["ragged"], np.concatenate([np.full((3 + 2 * k, 2), float(k)) for k in range(world)]))
        (f0, g0), (f1, g1) = o["fused"], o["sharded"]
        assert f1 == f0 and np.array_equal(g1, g0)         # e3 over ncclAllGather: bit for bit the fused evaluation
        m = o["model"]
        np.testing.assert_allclose(m["theta"], ref["theta"], rtol=1e-9)
        np.testing.assert_allclose(m["mean"], ref["mean"], rtol=0, atol=1e-7)
        np.testing.assert_allclose(m["var"], ref["var"], rtol=0, atol=1e-7)
        np.testing.assert_array_equal(m["mean"], out[0]["model"]["mean"])
        np.testing.assert_array_equal(m["acquired"], ref["acquired"])           # panel rows over RCCL: the same acquisitions
        assert m["evals"] < ref["evals"]
        # the sequential evaluations were SHARED (shard_sequential): the low-fidelity run by all ranks -- ONE optimiser, on rank 0,
        # whose steps are bit for bit the single process's (the sharded evaluation is) -- and, with two restarts on three ranks,
        # first run -> restart 0 by ranks 0 and 1
        assert (world, r) in m["groups"]
        np.testing.assert_array_equal(m["lf_theta"], ref["lf_theta"])
        assert m["lf_evals"] == (ref["lf_evals"] if r == 0 else 0)            # ONE optimiser: the followers served, they issued nothing
        m2 = o["model_r2"]
        np.testing.assert_array_equal(m2["theta"], ref_r2["theta"])
        np.testing.assert_allclose(m2["mean"], ref_r2["mean"], rtol=0, atol=1e-9)
        if world == 3:
            assert ((2, r) in m2["groups"]) == (r in (0, 1)), m2["groups"]
        for n, es in o["eval_sharded"].items():            # bitwise NLML / gradient (and predictions) on every rank
            (f0, g0, m0, v0), (f1, g1, m1, v1) = es["single"], es["sharded"]
            assert f1 == f0 and np.array_equal(g1, g0), (r, n)
            assert np.array_equal(m1, m0) and np.array_equal(v1, v0), (r, n)
            assert es["nograd"][0] == es["nograd"][1]
        if world == 3:
            # two groups per fit (the low-fidelity run's, the chain group's), formed ONCE for the two fits on the shared engines
            assert o["group_reuse"] == dict(formed=2, same_fit=True), o["group_reuse"]
        if world <= 3:
            for n in (900, 4096):                          # the distributed Cholesky: bitwise the single evaluation, factor included
                d = o["dist"][n]
                (f0, g0, m0, v0), (f1, g1, m1, v1) = d["single"], d["sharded"]
                assert f1 == f0 and np.array_equal(g1, g0) and d["L_equal"], (r, n)
                assert np.array_equal(m1, m0) and np.array_equal(v1, v0), (r, n)
            f2, g2, f3, g3 = o["dist"]["lead"]
            if r == 0:
                assert f2 == f3 and np.array_equal(g2, g3)
            if world == 2 and not dist_everywhere:
                f0, g0, f1, g1 = o["dist"]["measured_8192"]
                assert f1 == f0 and np.array_equal(g1, g0)
                cal, dec = o["dist"]["calibration"], o["dist"]["decision_8192"]
                assert cal is not None and cal["broadcast_us"] > 0 and cal["allgather_us"] > 0 and cal["reps"] == 20, cal
                # the rig's collectives go through RCCL's socket transport on loopback: far above what 127 of them may cost
                assert dec["cholesky"] == "replicated" and not dec["forced_by_MFGP_DIST_CHOL"], dec
                assert dec["collectives_on_chain"] == 64 - 1 + 16 and dec["measured_us_per_collective"] > 0, dec   # one all-gather per column + one broadcast per macro panel of 4
                assert dec["collective_cost_ms"] * 1.25 >= dec["projected_saving_ms"] and "collectives x" in dec["why"], dec
                if r == 0:
                    print("2-rank rig: calibration %s -> N = 8192: %s" % (cal, dec))
                if not isinstance(o["dist"]["forced_16384"], str):
                    f0, g0, f1, g1 = o["dist"]["forced_16384"]
                    assert f1 == f0 and np.array_equal(g1, g0)
                    assert o["dist"]["decision_16384"]["cholesky"] == "distributed" and o["dist"]["decision_16384"]["forced_by_MFGP_DIST_CHOL"]
            p_single, p_dist = o["dist"]["not_pd"]
            assert p_single > 0 and p_dist == p_single, o["dist"]["not_pd"]


@pytest.mark.parametrize("N", [300, 1500, 2100, 4200])
def test_sharded_evaluation_in_the_group_of_one_and_each_ranks_share(engine, N):
    """mfgp_eval_sharded without a communicator (the group of one) is mfgp_eval bit for bit -- single-macro plans and
    multi-macro ones -- and the device work of rank r of G alone (mfgp_dbg_eval_as_rank: no exchange) runs through for
    every (r, G) and leaves the handle without a factorisation."""
    rng = np.random.default_rng(N)
    X = rng.uniform(size=(N, 4))
    Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
    Y = cases.hf_4d(X)
    engine.set_data(Xa, Y)
    engine.set_kernel(cases.composite(4, 1))
    theta, noise = np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.01 * Y.var()
    f0, g0 = engine.eval(theta, noise, 1e-8)
    m0, v0 = engine.predict(Xa[:50])
    f1, g1 = engine.eval_sharded(theta, noise, 1e-8)
    m1, v1 = engine.predict(Xa[:50])
    assert f1 == f0 and np.array_equal(g1, g0) and np.array_equal(m1, m0) and np.array_equal(v1, v0)
    for G in (2, 3, 8):
        for r in (0, G - 1):
            assert engine.dbg_eval_as_rank(theta, noise, r, G) > 0.0
    with pytest.raises(RuntimeError):
        engine.predict(Xa[:5])                             # a lone rank's share is not a factorisation
    f2, g2 = engine.eval(theta, noise, 1e-8)               # the handle is intact
    assert f2 == f0 and np.array_equal(g2, g0)


def test_rccl_calls_with_a_communicator_of_one(engine):
    """mfgp_comm_unique_id / mfgp_comm_init / mfgp_allgather_rows / mfgp_allgather_host on the real RCCL, world size 1
    (all a one-GPU box admits): the library loads librccl lazily, creates the communicator on the handle's device and
    runs both collectives on the handle's stream."""
    from multifidelity_datafusion_gps_amd.sharding import SocketComm, eval_rowblock_allgather
    rng = np.random.default_rng(5)
    X = rng.uniform(size=(500, 3))
    Y = cases.hf_3d(X)
    engine.set_data(X, Y)
    engine.set_kernel(cases.single(cases.RBF, 3))
    theta, noise = np.array([1.0, 0.3]), 0.01
    f0, g0 = engine.eval(theta, noise, 1e-8)
    comm = SocketComm(0, 1)
    assert comm.attach_engine(engine) and comm.transport == "rccl" and engine.comm_size == 1
    got = engine.allgather_host(np.arange(10.0))
    assert got.shape == (1, 10) and np.array_equal(got[0], np.arange(10.0))
    engine.kbuild_rows(theta, noise, 1e-8, 0, 512)
    engine.allgather_rows()                                       # in place, one rank: the matrix is unchanged
    f1, g1 = engine.eval_prebuilt(True)
    assert f1 == f0 and np.array_equal(g1, g0)
    f2, g2 = eval_rowblock_allgather(engine, comm, theta, noise)
    assert f2 == f0 and np.array_equal(g2, g0)
    engine.comm_destroy()


def _failing_leader_worker(rank, port, q):
    import os
    import sys
    import time
    from multifidelity_datafusion_gps_amd import sharding
    os.environ.update(sharding.rehearsal_env(rank))
    os.environ["MFGP_SHARD_TIMEOUT_S"] = "3"               # a follower whose leader is gone gives up after this long
    from multifidelity_datafusion_gps_amd import engine as gp
    from multifidelity_datafusion_gps_amd._lib import Engine
    comm = sharding.SocketComm(rank, 2, "127.0.0.1", port, timeout=120)
    e = Engine(0)
    comm.attach_engine(e, required=True, init_timeout=90)
    rng = np.random.default_rng(5)
    X = rng.uniform(size=(700, 2))
    m = gp.GPRegression(X, cases.hf_2d(X)[:, None], kernel=gp.RBF(2), engine=e)
    group = comm.shard_group(e, [0, 1])
    assert group is not None and group.size == 2
    comm.barrier()
    t0 = time.perf_counter()
    res = {"rank": rank}
    try:
        if group.leads:
            e.dbg_fail_sharded_after(3)                    # the third evaluation of the run fails AFTER the followers were told to start it
        group.run(m, (lambda: m.optimize(max_iters=10)) if group.leads else None)
        res["error"] = None
    except RuntimeError as ex:
        res["error"] = str(ex)
    res["seconds"] = time.perf_counter() - t0
    res["aborted"] = bool(e.comm_aborted)
    res["evals"] = int(m.n_evals)
    if group.leads:
        try:                                               # a further collective on the poisoned handle is REFUSED, not enqueued
            e.sharded_release()
            res["release"] = "issued"
        except RuntimeError as ex:
            res["release"] = str(ex)
    q.put((rank, res))
    q.close()
    q.join_thread()                                        # (the queue's feeder thread must have written the item before the hard exit)
    sys.stderr.write("rank %d: %s\n" % (rank, res))
    sys.stderr.flush()
    os._exit(7 if res["error"] else 0)                     # (the rank ends with an error: its launcher would stop the peers)


def test_a_leader_that_fails_inside_a_shared_pass_ends_the_group_instead_of_hanging_it():
    """the leader of a shared evaluation fails after the control block told the follower to start the pass.  It must
    not issue another collective (the release would never be matched): its communicator is aborted, ShardGroup.run skips the
    release, the exception ends the rank -- and the follower, left inside the pass, gives up after MFGP_SHARD_TIMEOUT_S with an
    error of its own instead of waiting for ever."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_leader_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        out = dict(q.get(timeout=150) for _ in procs)
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    lead, foll = out[0], out[1]
    assert "injected failure" in lead["error"] and "communicator was aborted" in lead["error"], lead
    assert lead["aborted"] and lead["evals"] <= 3 and lead["seconds"] < 30, lead
    assert "aborted" in lead["release"] and lead["release"] != "issued", lead
    assert foll["error"] is not None and foll["aborted"] and foll["seconds"] < 60, foll
    assert [p.exitcode for p in procs] == [7, 7]


def _unmatched_gather_worker(rank, port, q, mode="absent_peer"):
    import os
    import sys
    import time
    from multifidelity_datafusion_gps_amd import sharding
    os.environ.update(sharding.rehearsal_env(rank))
    os.environ["MFGP_SHARD_TIMEOUT_S"] = "3"
    from multifidelity_datafusion_gps_amd._lib import Engine
    comm = sharding.SocketComm(rank, 2, "127.0.0.1", port, timeout=120)
    e = Engine(0)
    comm.attach_engine(e, required=True, init_timeout=90)
    np.testing.assert_array_equal(e.allgather_host(np.full(3, float(rank))).reshape(-1), np.repeat([0.0, 1.0], 3))    # a matched gather first
    rows = mode == "rccl_error_rows"
    if rows:
        rng = np.random.default_rng(5)
        X = rng.uniform(size=(700, 3))
        e.set_data(X, cases.hf_3d(X)); e.set_kernel(cases.single(cases.RBF, 3))
        e.kbuild_owned_rows(np.array([1.0, 0.4]), 0.01, 1e-8, rank, 2)
    comm.barrier()
    res = {"rank": rank, "error": None}
    t0 = time.perf_counter()
    gather = (lambda: e.allgather_rows()) if rows else (lambda: e.allgather_host(np.arange(5.0)))
    if rank == 0 or mode != "absent_peer":
        if rank == 0 and mode != "absent_peer":
            e.dbg_fail_collective_after(1)                 # this rank's ncclAllGather "returns an error": the peer's is never matched
        try:
            gather()                                       # absent_peer: rank 1 never issues its half
        except RuntimeError as ex:
            res["error"] = str(ex)
        res["seconds"] = time.perf_counter() - t0
        res["aborted"] = bool(e.comm_aborted)
        try:
            e.allgather_host(np.arange(5.0))
            res["again"] = "issued"
        except RuntimeError as ex:
            res["again"] = str(ex)
    else:
        time.sleep(7.0)                                    # alive, its communicator intact, but somewhere else in the protocol
    q.put((rank, res))
    q.close()
    q.join_thread()
    sys.stderr.write("rank %d: %s\n" % (rank, res))
    sys.stderr.flush()
    os._exit(0)


def _run_gather_pair(mode):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_unmatched_gather_worker, args=(r, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        out = dict(q.get(timeout=150) for _ in procs)
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    return out


def test_a_gather_the_peer_never_joins_gives_up_after_the_deadline():
    """Every wait for a stream that carries a collective is the poll with a deadline (comm_stream_wait), the small host gathers and
    the row-block gather included: a peer that never issues its half costs MFGP_SHARD_TIMEOUT_S, then the communicator is aborted,
    the call fails with a message that says so, and further collectives on the handle are refused."""
    out = _run_gather_pair("absent_peer")
    lone = out[0]
    assert lone["error"] is not None and "no progress for 3 s" in lone["error"], lone
    assert lone["aborted"] and 2.5 < lone["seconds"] < 30.0, lone
    assert lone["again"] != "issued" and "aborted" in lone["again"], lone


@pytest.mark.parametrize("mode", ["rccl_error_host", "rccl_error_rows"])
def test_a_gather_whose_rccl_call_fails_aborts_the_communicator_and_frees_the_peer(mode):
    """an ncclAllGather that comes back with an error (injected: mfgp_dbg_fail_collective_after) used to return -4
    with the communicator intact -- the peer then sat in its half of the collective until ITS deadline with nothing telling the
    failed rank's later calls to stay away.  Now the failing rank aborts its communicator at once (state -1, further collectives
    refused, what it had enqueued drained), and the peer is out of the collective within its deadline with the same state."""
    out = _run_gather_pair(mode)
    failed, peer = out[0], out[1]
    assert failed["error"] is not None and "communicator was aborted" in failed["error"], failed
    assert failed["aborted"] and failed["seconds"] < 3.0, failed
    assert failed["again"] != "issued" and "aborted" in failed["again"], failed
    assert peer["error"] is not None and "no progress for 3 s" in peer["error"], peer
    assert peer["aborted"] and 2.5 < peer["seconds"] < 30.0, peer
    assert peer["again"] != "issued" and "aborted" in peer["again"], peer
