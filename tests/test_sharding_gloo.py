"""world_size-2 gloo test of the N > 1 path: restarts and predictive rows sharded over ranks
give exactly the single-process result.  The engine is the tests-only oracle double; the sharding logic under
test is the product's (engine.GPRegression.optimize_restarts, MultifidelityDataFusion.predict, sharding.TorchComm)."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest


def hf(x):
    return (np.sin(2.2 * np.pi * x[:, 0]) * np.sin(np.pi * x[:, 1]))[:, None]


def lf(x):
    return hf(x) - 1.2 * (np.sin(x[:, :1] * np.pi * 0.1) + np.sin(x[:, 1:2] * np.pi * 0.1))


def _run_model(comm, conc=1):
    import multifidelity_datafusion_gps_amd as mf
    from tests.oracle_engine import OracleEngine
    rng = np.random.default_rng(7)
    X_hf = rng.uniform(size=(24, 2))
    X_st = rng.uniform(size=(41, 2))
    engines = {"lf": OracleEngine(), "hf": OracleEngine(), "hf#1": OracleEngine(), "hf#2": OracleEngine()}
    model = mf.NARGP(2, hf, lf, seed=11, comm=comm, engines=engines)
    model.first_run_max_iters, model.restart_max_iters, model.restart_concurrency = 25, 25, conc
    model.fit(X_hf)
    mean, var = model.predict(X_st)
    theta = np.array([p.value for p in model.hf_model.parameters()])
    return dict(theta=theta, mean=mean, var=var, evals=sum(e.n_evals for k, e in engines.items() if k != "lf"))


def _worker(rank, world, port, q, conc=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from multifidelity_datafusion_gps_amd.sharding import TorchComm
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = TorchComm()
        assert comm.size == world and comm.rank == rank
        # ragged row all-gather
        rows = np.full((rank + 2, 3), float(rank))
        g = comm.allgather_rows(rows)
        assert g.shape == (2 + 3, 3) and g[:2].sum() == 0 and g[2:].sum() == 9
        q.put((rank, _run_model(comm, conc)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("conc", [1, 2])
def test_two_rank_fit_predict_equals_single_process(conc):
    """conc = 1: sequential recipe, restarts round-robin over the ranks.  conc = 2: the randomized restarts run in
    background threads beside the first run, rank 0 keeps the sequential first run -> restart 0 and the others are
    balanced over the ranks (AbstractMFGP.assign_restarts) -- same runs, same winner."""
    from multifidelity_datafusion_gps_amd.sharding import LocalComm
    ref = _run_model(LocalComm(), conc)
    if conc == 2:
        seq = _run_model(LocalComm(), 1)
        np.testing.assert_allclose(ref["theta"], seq["theta"], rtol=1e-12)     # concurrency does not change the fit
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, conc)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        np.testing.assert_allclose(out[r]["theta"], ref["theta"], rtol=1e-12)   # same winner on every rank
        # row blocks go through differently shaped LAPACK/BLAS calls: equal up to rounding amplified by cond(Ky)
        np.testing.assert_allclose(out[r]["mean"], ref["mean"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(out[r]["var"], ref["var"], rtol=0, atol=1e-6)
        np.testing.assert_array_equal(out[r]["mean"], out[0]["mean"])            # every rank holds the same gathered result
    # the 6 restarts were split: each rank issued fewer evaluations than the single process
    assert out[0]["evals"] < ref["evals"] and out[1]["evals"] < ref["evals"]
