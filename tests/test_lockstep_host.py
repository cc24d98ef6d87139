"""The lock-stepped restarts on the CPU (the double with eval_batch, tests/oracle_engine.py::BatchOracleEngine): the recipe's 1 + 6
L-BFGS-B runs (src/abstractMFGP.py:131-137) as generators over scipy's L-BFGS-B core by reverse communication (lbfgsb.py,
engine.LockstepLane), as threads over scipy's blocking fmin_l_bfgs_b (engine.LockstepEvaluator), and in the reference's sequential
order -- the same runs, the same evaluations, the same fitted parameters, bit for bit.  (On the GPU: tests/test_gpu_models.py.)"""
import numpy as np
import pytest

import multifidelity_datafusion_gps_amd as mf
from multifidelity_datafusion_gps_amd import engine as gp
from multifidelity_datafusion_gps_amd import lbfgsb
from tests import cases
from tests.oracle_engine import BatchOracleEngine, OracleEngine


def col(f):
    return lambda x: f(x)[:, None]


def test_reverse_communication_driver_is_fmin_l_bfgs_b():
    """lbfgsb.Lbfgsb against scipy.optimize.fmin_l_bfgs_b on GP objectives: same points asked for, same end, same counts and status"""
    from scipy.optimize import fmin_l_bfgs_b
    assert lbfgsb.available()
    rng = np.random.default_rng(0)
    X = rng.uniform(size=(40, 3))
    Y = (np.sin(6.0 * X[:, 0]) + 0.1 * X[:, 1])[:, None]
    for kern, budget in ((gp.RBF(3), 300), (gp.RBF(3, ARD=True), 300), (gp.Matern52(3), 300), (gp.RBF(3, ARD=True), 17), (gp.Matern32(3), 5)):
        m = gp.GPRegression(X, Y, kernel=kern, engine=OracleEngine())
        x0 = m.optimizer_array.copy()
        seen_a, seen_b = [], []

        def fa(x):
            seen_a.append(np.array(x))
            return m._objective_grads(x)

        def fb(x):
            seen_b.append(np.array(x))
            return m._objective_grads(x)
        xa, va, da = fmin_l_bfgs_b(fa, x0, maxfun=budget, maxiter=budget)
        xb, vb, db = lbfgsb.minimize(fb, x0, maxfun=budget, maxiter=budget)
        assert np.array_equal(xa, xb) and va == vb
        assert (da["funcalls"], da["nit"], da["warnflag"], da["task"]) == (db["funcalls"], db["nit"], db["warnflag"], db["task"])
        assert len(seen_a) == len(seen_b) and all(np.array_equal(p, q) for p, q in zip(seen_a, seen_b))


def _fit(mode, n_hf=40, evals=25, cap=None, seed=3):
    kw = {"sequential": dict(restart_lockstep=False, restart_concurrency=1),
          "programs, 1 lane": dict(restart_lockstep=True, lockstep_lanes=1),
          "programs, 2 lanes": dict(restart_lockstep=True, lockstep_lanes=2),
          "programs, 3 lanes, 4 live": dict(restart_lockstep=True, lockstep_lanes=3, lockstep_width=4),
          "threads, 2 lanes": dict(restart_lockstep=True, lockstep_lanes=2, lockstep_threads=True)}[mode]
    M = type("M", (mf.NARGP,), dict(first_run_max_iters=evals, restart_max_iters=evals, eval_cap=cap, **kw))
    engines = {k: BatchOracleEngine() for k in ("hf", "lf", "hf#1", "hf#2")}
    m = M(2, col(cases.hf_2d), col(cases.lf_2d), seed=seed, engines=engines)
    rng = np.random.default_rng(1)
    m.fit(rng.uniform(size=(n_hf, 2)))
    theta = np.array([p.value for p in m.hf_model.parameters()])
    runs = sorted((round(r.f_opt, 12), tuple(np.round(r.x_opt, 12))) for r in m.hf_model.optimization_runs)
    mean, var = m.predict(rng.uniform(size=(16, 2)))
    lanes = getattr(m, "last_lockstep_lanes", None)
    return theta, runs, m.hf_model.n_evals, mean, var, lanes, sum(r.n_evals for r in m.hf_model.optimization_runs)


@pytest.mark.parametrize("cap", [None, 12])
def test_lockstep_programs_threads_and_the_sequential_order_give_the_same_fit(cap):
    ref = _fit("sequential", cap=cap)
    assert len(ref[1]) == 7                                    # first run + restart 0 + five randomized restarts
    for mode in ("programs, 1 lane", "programs, 2 lanes", "programs, 3 lanes, 4 live", "threads, 2 lanes"):
        got = _fit(mode, cap=cap)
        assert np.array_equal(got[0], ref[0]), mode            # fitted hyper-parameters, bit for bit
        assert got[1] == ref[1], mode                          # every run ended where it ends alone
        assert got[2] == ref[2], mode                          # and took the evaluations it takes alone
        assert np.array_equal(got[3], ref[3]) and np.array_equal(got[4], ref[4]), mode
        lanes = got[5]
        # every evaluation of the seven runs went through a lane's batched passes (the model's own count has one more: the
        # factorisation at the winner that the fit ends with)
        assert sum(ln.evals for ln in lanes) == got[6] == ref[6] == ref[2] - 1, mode
        assert all(max(ln.round_sizes) <= 6 for ln in lanes if ln.round_sizes), mode
    one = _fit("programs, 1 lane", cap=cap)[5]
    assert len(one) == 1 and max(one[0].round_sizes) == 6      # the sequential pair + five restarts in one pass per round


def test_lockstep_falls_back_to_threads_without_the_private_core(monkeypatch):
    ref = _fit("programs, 2 lanes")
    monkeypatch.setattr(lbfgsb, "_checked", False)             # as if scipy's core could not be driven by reverse communication
    got = _fit("programs, 2 lanes")
    assert isinstance(got[5][0], gp.LockstepEvaluator)
    assert np.array_equal(got[0], ref[0]) and got[1] == ref[1] and got[2] == ref[2]


def test_a_failing_engine_is_a_failed_evaluation_on_every_path():
    """an engine that raises (the CPU double's "not positive definite, even with jitter"; a ValueError) is a FAILED evaluation --
    DBL_MAX and the previous gradient, up to paramz' ten in a row -- whether the run goes through the model's own objective, the
    run generator on one handle, or a lane's batched passes; and all three take the same steps"""
    class Flaky(BatchOracleEngine):
        fail_at = (4, 11)

        def eval(self, theta, noise, jitter=1e-8, want_grad=True):
            if self.n_evals + 1 in self.fail_at:
                self.n_evals += 1
                raise np.linalg.LinAlgError("not positive definite, even with jitter.")
            return super().eval(theta, noise, jitter, want_grad)

        def eval_batch(self, thetas, noises, jitters=1e-8, want_grad=True):
            if self.n_evals + 1 in self.fail_at:
                self.n_evals += len(np.atleast_2d(thetas))
                raise ValueError("engine refused the batch")
            return super().eval_batch(thetas, noises, jitters, want_grad)

    rng = np.random.default_rng(0)
    X = rng.uniform(size=(30, 3))
    Y = (np.sin(5.0 * X[:, :1]) + X[:, 1:2] * X[:, 2:])
    ends = {}
    for fast in (True, False):
        m = gp.GPRegression(X, Y, kernel=gp.Matern52(3, ARD=True), engine=Flaky())
        m.fast_optimize = fast
        run = m.optimize(max_iters=40)
        ends[fast] = (run.x_opt, run.f_opt, m.n_evals, run.n_evals, run.status)
        assert m._engine.n_evals > m.n_evals and run.f_opt < 1e300   # the engine failed once, nothing escaped, the run has an optimum
    assert np.array_equal(ends[True][0], ends[False][0]) and ends[True][1:] == ends[False][1:]
    m = gp.GPRegression(X, Y, kernel=gp.Matern52(3, ARD=True), engine=Flaky())
    lane = gp.LockstepLane(m._engine)
    got = []
    lane.drive([m.optimize_program(40, got)])
    assert np.array_equal(got[0].x_opt, ends[True][0]) and got[0].f_opt == ends[True][1]


def test_reverse_communication_driver_on_random_functions_and_budgets():
    """300 seeded random problems -- dimension 1 .. 12, budgets 1 .. 60, smooth bowls, Rosenbrock chains, functions that now and then
    answer DBL_MAX with a stale gradient (a failed GP evaluation) or NaN: the ask / tell driver and fmin_l_bfgs_b ask for the same points
    and end the same way, bit for bit"""
    from scipy.optimize import fmin_l_bfgs_b
    rng = np.random.default_rng(2024)
    for case in range(300):
        n = int(rng.integers(1, 13))
        budget = int(rng.integers(1, 61))
        kind = int(rng.integers(0, 4))
        A = rng.standard_normal((n, n))
        A = A @ A.T + 0.1 * np.eye(n)
        b = rng.standard_normal(n)
        bad_every = int(rng.integers(3, 9))
        x0 = rng.standard_normal(n) * 2.0

        def make():
            seen = []

            def f(x):
                seen.append(np.array(x))
                if kind == 0:
                    return float(0.5 * x @ A @ x - b @ x), A @ x - b
                if kind == 1 and n >= 2:
                    a_, b_ = x[:-1], x[1:]
                    g = np.zeros_like(x)
                    g[:-1] += -400.0 * a_ * (b_ - a_ * a_) - 2.0 * (1.0 - a_)
                    g[1:] += 200.0 * (b_ - a_ * a_)
                    return float(np.sum(100.0 * (b_ - a_ * a_) ** 2 + (1.0 - a_) ** 2)), g
                val, g = float(np.sum(np.cosh(x)) + 0.5 * x @ A @ x), np.sinh(x) + A @ x
                if len(seen) % bad_every == 0:
                    return (np.finfo(float).max, np.clip(g, -1e10, 1e10)) if kind == 2 else (float("nan"), g)
                return val, g
            return f, seen
        fa, sa = make()
        fb, sb = make()
        xa, va, da = fmin_l_bfgs_b(fa, x0, maxfun=budget, maxiter=budget)
        xb, vb, db = lbfgsb.minimize(fb, x0, maxfun=budget, maxiter=budget)
        same_f = (va == vb) or (np.isnan(va) and np.isnan(vb))
        assert np.array_equal(xa, xb, equal_nan=True) and same_f, (case, n, budget, kind)
        assert (da["funcalls"], da["nit"], da["warnflag"], da["task"]) == (db["funcalls"], db["nit"], db["warnflag"], db["task"]), (case, kind)
        assert len(sa) == len(sb) and all(np.array_equal(p, q, equal_nan=True) for p, q in zip(sa, sb)), (case, kind)


class _TightEngine(BatchOracleEngine):
    """the CPU double with a device-memory budget: eval_batch raises EngineOutOfMemory for more sets than `sets_that_fit`; mem_info /
    batch_mem answer what the HIP engine's would (1 byte per set)"""
    sets_that_fit = 16
    advertised_free = 1 << 40
    MAX_BATCH = 16

    def __init__(self):
        super().__init__()
        self.passes = []

    def eval_batch(self, thetas, noises, jitters=1e-8, want_grad=True):
        from multifidelity_datafusion_gps_amd._lib import EngineOutOfMemory
        B = len(np.atleast_2d(thetas))
        if B > self.sets_that_fit:
            raise EngineOutOfMemory("mfgp_eval_batch: out of device memory for %d matrix sets" % B)
        self.passes.append(B)
        return super().eval_batch(thetas, noises, jitters, want_grad)

    def eval(self, theta, noise, jitter=1e-8, want_grad=True):
        self.passes.append(0)
        return super().eval(theta, noise, jitter, want_grad)

    def mem_info(self):
        return self.advertised_free, 1 << 41

    def batch_mem(self, sets):
        return int(sets), 0, 0


def _tight_fit(fit, free, width=None):
    E = type("E", (_TightEngine,), dict(sets_that_fit=fit, advertised_free=free))
    M = type("M", (mf.NARGP,), dict(first_run_max_iters=25, restart_max_iters=25, restart_lockstep=True, lockstep_lanes=1,
                                    lockstep_width=width, memory_reserve=staticmethod(lambda e: 0)))
    engines = {k: E() for k in ("hf", "lf")}
    m = M(2, col(cases.hf_2d), col(cases.lf_2d), seed=3, engines=engines)
    m.fit(np.random.default_rng(1).uniform(size=(40, 2)))
    theta = np.array([p.value for p in m.hf_model.parameters()])
    runs = sorted((round(r.f_opt, 12), tuple(np.round(r.x_opt, 12))) for r in m.hf_model.optimization_runs)
    return theta, runs, m.hf_model.n_evals, m.last_fit_info, [p for p in engines["hf"].passes]


def test_batch_memory_policy_narrows_the_passes_and_keeps_the_fit():
    """the width of a batched pass follows the device's free memory -- sized BEFORE asking (mem_info / batch_mem),
    halved WHEN the engine answers out-of-memory anyway, request by request at the end -- and the fit does not change by a bit."""
    ref = _tight_fit(16, 1 << 40)
    assert ref[3]["driver"] == "lbfgsb-generators" and ref[3]["sets_per_pass"] == [6] and ref[3]["oom_fallbacks"] == [[]]
    assert max(ref[4]) == 6
    # (1) sized from what the device reports free: 2 sets, then none (single evaluations on the handle's own slab)
    for free, widest in ((2, 2), (0, 0)):
        got = _tight_fit(16, free)
        assert got[3]["sets_per_pass"] == [free] and got[3]["oom_fallbacks"] == [[]]
        assert max(got[4][1:]) == widest, (free, sorted(set(got[4])))      # ([0]: the LF level's own evaluations come first)
        assert np.array_equal(got[0], ref[0]) and got[1] == ref[1] and got[2] == ref[2]
    # (2) the device reports room but the allocation fails (another process took it): 6 -> 3 -> 2 -> 1 sets on out-of-memory answers
    for fit, used, steps in ((3, 3, [(6, 3)]), (2, 2, [(6, 3), (3, 2)]), (0, 1, [(6, 3), (3, 2), (2, 1)])):
        got = _tight_fit(fit, 1 << 40)
        assert got[3]["sets_per_pass"] == [6] and got[3]["sets_per_pass_used"] == [used], got[3]
        assert got[3]["oom_fallbacks"][0][:len(steps)] == steps, got[3]
        assert np.array_equal(got[0], ref[0]) and got[1] == ref[1] and got[2] == ref[2]


def test_failed_evaluation_state_persists_across_a_models_sequential_runs():
    """paramz keeps the count of consecutive failed evaluations and the last good gradient ON THE MODEL, across the
    optimize() calls of a fit.  A second optimize() whose FIRST evaluation fails is answered with the first run's last good gradient
    (not zeros), on the fast path (run generator) exactly as through the model's own objective."""
    class FailsOnce(BatchOracleEngine):
        fail_eval = None

        def eval(self, theta, noise, jitter=1e-8, want_grad=True):
            if self.n_evals + 1 == self.fail_eval:
                self.n_evals += 1
                raise ValueError("engine refused the evaluation")
            return super().eval(theta, noise, jitter, want_grad)

    rng = np.random.default_rng(2)
    X = rng.uniform(size=(30, 2))
    Y = np.sin(5.0 * X[:, :1]) * X[:, 1:2]
    ends = {}
    for fast in (True, False):
        m = gp.GPRegression(X, Y, kernel=gp.RBF(2, ARD=True), engine=FailsOnce())
        m.fast_optimize = fast
        r1 = m.optimize(max_iters=15)
        good = np.array(m._last_good_grad)
        m._engine.fail_eval = m._engine.n_evals + 1          # the first evaluation of the next run
        seen = []
        orig = gp._lbfgsb.Lbfgsb.tell

        def tell(self_, f, g, _seen=seen, _orig=orig):
            _seen.append((float(f), np.array(g)))
            return _orig(self_, f, g)
        gp._lbfgsb.Lbfgsb.tell = tell
        try:
            r2 = m.optimize(max_iters=15)
        finally:
            gp._lbfgsb.Lbfgsb.tell = orig
        assert seen[0][0] == gp._F_FAILED and np.array_equal(seen[0][1], good), fast      # the previous run's gradient, not zeros
        assert m.n_failed_evals == 1 and m._fail_count == 0
        ends[fast] = (r1.x_opt, r1.f_opt, r2.x_opt, r2.f_opt, r2.n_evals, m.n_evals)
    assert all(np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b for a, b in zip(ends[True], ends[False]))


def test_thread_per_run_lockstep_survives_out_of_memory_batches():
    """the thread-per-run form (engine.LockstepEvaluator: the fallback for a scipy whose L-BFGS-B core cannot be driven by reverse
    communication) meets an engine whose batches never fit: every round goes request by request on the handle's own slab, the fit is
    the sequential order's bit for bit, and the lanes say how many rounds fell back"""
    ref = _fit("sequential")
    E = type("E", (_TightEngine,), dict(sets_that_fit=0, advertised_free=1 << 40))
    M = type("M", (mf.NARGP,), dict(first_run_max_iters=25, restart_max_iters=25, restart_lockstep=True, lockstep_lanes=2,
                                    lockstep_threads=True))
    engines = {k: E() for k in ("hf", "lf", "hf#1")}
    m = M(2, col(cases.hf_2d), col(cases.lf_2d), seed=3, engines=engines)
    rng = np.random.default_rng(1)
    m.fit(rng.uniform(size=(40, 2)))
    theta = np.array([p.value for p in m.hf_model.parameters()])
    runs = sorted((round(r.f_opt, 12), tuple(np.round(r.x_opt, 12))) for r in m.hf_model.optimization_runs)
    mean, var = m.predict(rng.uniform(size=(16, 2)))           # (the factorisation at the winner: the model's count includes it)
    assert np.array_equal(theta, ref[0]) and runs == ref[1] and m.hf_model.n_evals == ref[2]
    assert np.array_equal(mean, ref[3]) and np.array_equal(var, ref[4])
    lanes = m.last_lockstep_lanes
    assert all(isinstance(ln, gp.LockstepEvaluator) for ln in lanes)
    assert sum(ln.oom_fallbacks for ln in lanes) == sum(1 for ln in lanes for n in ln.round_sizes if n >= 1) > 0
    assert m.last_fit_info["driver"].startswith("thread-per-run")
    assert all(p == 0 for p in engines["hf"].passes)            # no batch ever ran: single evaluations only
