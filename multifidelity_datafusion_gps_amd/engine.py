"""GPRegression-shaped host objects over the HIP engine: the drop-in for the slice of the GPy API the
reference touches.

What the reference calls, and where (all under /root/reference):
  * GPy.kern.RBF(input_dim, active_dims=...), k*k, k+k                 src/abstractMFGP.py:60,77-80
  * GPy.models.GPRegression(X=, Y=, kernel=, initialize=True)          src/MFDataFusion.py:93-98, src/abstractMFGP.py:100-102
  * model[".*Gaussian_noise"] = v / .fix() / .unfix() / .constrain_positive(), model.Y
                                                                       src/abstractMFGP.py:132-136
  * model.optimize(max_iters=), model.optimize_restarts(n, optimizer="bfgs", max_iters=, verbose=)
                                                                       src/abstractMFGP.py:103,134,137
  * model.likelihood.variance = 1e-6                                   src/MFDataFusion.py:155
  * model.predict(X*) -> (mean (N*,1), variance incl. noise (N*,1))     src/MFDataFusion.py:156, src/abstractMFGP.py:104
  * kernel.to_dict()["parts"]...["lengthscale"]                        src/models/GPDFC.py:26-29

Semantics restated from GPy 1.9.9 / paramz 0.9.5 (not vendored; statements tagged [GPy-recall]):
positive parameters live in a softplus ("Logexp") optimizer space; the objective is the negative log
marginal likelihood; L-BFGS-B (scipy fmin_l_bfgs_b, max_iters -> maxfun and maxiter) drives it;
optimize_restarts keeps restart 0 at the current point, randomizes the others with N(0,1) draws in
optimizer space and installs the best; a failed Cholesky is retried with jitter mean(diag)*1e-6*10^k.

All arithmetic (K build, Cholesky, solves, log-det, gradient, prediction) runs in libmfgp_hip.so on
the GPU through _lib.Engine.  There is no CPU fallback.  The evaluation is LAZY: changing a parameter
only marks the model dirty; the factorisation happens at the next query (GPy re-runs inference eagerly
on every assignment -- same results, fewer O(N^3) passes).
"""
import re
import time

import numpy as np
from scipy import optimize as _sciopt

from . import _lib
from . import lbfgsb as _lbfgsb
from ._lib import KERN_ARD, KERN_MATERN32, KERN_MATERN52, KERN_RBF, NotPositiveDefinite
from .kern import *  # noqa: F401,F403  (Param, RBF, Matern32, Matern52, Prod, Add, Gaussian, ...: this module stays the one import of the L3 code)
from .kern import (_LIM_VAL, _LOG_LIM_VAL, _ParamSelection, _ParamVector, _Combination, _logexp_f, _logexp_finv,  # noqa: F401
                   _logexp_gradfactor)
from .lockstep import *  # noqa: F401,F403
from .lockstep import (_BudgetExhausted, _EVAL_ERRORS, _F_FAILED, _G_CLIP_FAILED, _OptRun, _capped, _check_parameters)  # noqa: F401



class GPRegression:
    """Exact GP regression with a Gaussian likelihood on the HIP engine (GPy.models.GPRegression stand-in)."""

    _allowed_failures = 10  # paramz tolerates this many failed objective evaluations per model [GPy-recall]

    eval_cap = None   # hard cap on objective evaluations per optimize() / restart (None: scipy's maxfun semantics only)
    fast_optimize = True   # optimize() through the run generator (False: through the model's own Param objects, evaluation by evaluation)

    def __init__(self, X, Y, kernel=None, noise_var=1.0, initialize=True, engine=None, name="GP regression"):
        X = np.ascontiguousarray(X, dtype=np.float64)
        Y = np.ascontiguousarray(Y, dtype=np.float64)
        if X.ndim != 2:
            raise ValueError("X must be (N, D)")
        if Y.ndim == 1:
            Y = Y[:, None]
        if Y.shape != (X.shape[0], 1):
            raise ValueError("Y must be (N, 1)")
        self.name = name
        self.X, self.Y = X, Y
        self.kern = kernel if kernel is not None else RBF(X.shape[1])
        self.likelihood = Gaussian(noise_var, owner=self)
        self.Gaussian_noise = self.likelihood
        self.kern._set_owner(self)
        self._engine = engine if engine is not None else _lib.Engine()
        self._owns_engine = engine is None
        self._parts, self._part_params = self.kern.engine_parts()
        self._engine.set_data(self.X, self.Y[:, 0])
        self._engine.set_kernel(self._parts)
        self._dirty = True
        self._have_grad = False
        self._nlml = None
        self._grad_nat = None
        self._jitter_used = CONST_JITTER
        self._fail_count = 0
        self.optimization_runs = []
        self._eval_hook = None     # (theta, noise, jitter) -> (nlml, grad): evaluations routed through a LockstepEvaluator
        self.n_evals = 0           # objective(+gradient) evaluations issued to the GPU
        self.n_jitter_retries = 0  # ... of those, repeats of an evaluation whose Ky was not positive definite (jitchol's retries)
        self.n_failed_evals = 0    # evaluations that ended as FAILED (DBL_MAX + the previous gradient: paramz' convention)
        import threading
        self._evals_lock = threading.Lock()   # n_evals is bumped by every thread that evaluates for this model (lanes, background restarts)
        self._main_evals = 0       # ... of those, the ones issued through this object's own state (not by background restarts)
        self.update_model = True

    # ---- parameter plumbing ----------------------------------------------------------------------
    def _param_changed(self, p):
        self._dirty = True

    def parameters(self):
        return self.kern.parameters() + [self.likelihood.variance]

    def _named_parameters(self):
        out = []
        for i, (v, ls) in enumerate(self._part_params):
            for p, n in [(v, "variance")] + [(l, l.name) for l in ls]:
                if not any(p is q for _, q in out):
                    out.append(("%s.kern_%d.%s" % (self.name, i, n), p))
        out.append(("%s.Gaussian_noise.variance" % self.name, self.likelihood.variance))
        return out

    def _match(self, pattern):
        rx = re.compile(pattern)
        hits = [p for n, p in self._named_parameters() if rx.search(n)]
        if not hits:
            raise KeyError("no parameter matches %r" % pattern)
        return hits

    def __getitem__(self, pattern):
        return _ParamSelection(self._match(pattern))

    def __setitem__(self, pattern, value):
        for p in self._match(pattern):
            p.value = value

    # named equivalents (SURVEY 8(b) recommends these for fresh callers)
    def set_noise(self, v):
        self.likelihood.variance = v

    def fix_noise(self):
        self.likelihood.variance.fix()

    def unfix_noise(self):
        self.likelihood.variance.unfix()

    def _free_params(self):
        return [p for p in self.parameters() if not p.fixed]

    @property
    def optimizer_array(self):
        return _logexp_finv(np.array([p.value for p in self._free_params()]))

    @optimizer_array.setter
    def optimizer_array(self, x):
        vals = _logexp_f(np.asarray(x, dtype=np.float64))
        for p, v in zip(self._free_params(), vals):
            p._value = float(v)
        self._dirty = True

    def randomize(self, rand_gen=None):
        """N(0,1) draws in optimizer space for every free parameter [GPy-recall: paramz randomize]."""
        n = len(self._free_params())
        x = rand_gen(size=n) if rand_gen is not None else np.random.normal(size=n)
        self.optimizer_array = x

    # ---- evaluation --------------------------------------------------------------------------------
    def _theta(self):
        th = []
        for v, ls in self._part_params:
            th += [v.value] + [l.value for l in ls]
        return np.array(th)

    def _ensure(self, want_grad):
        """(re)factorise if parameters changed; GPy's jitchol retry policy on failure."""
        if not self._dirty and (self._have_grad or not want_grad):
            return
        theta, noise = self._theta(), self.likelihood.variance.value
        _check_parameters(theta, noise)
        if not self._dirty and want_grad:
            self._grad_nat = self._engine.nlml_grad()
            self._have_grad = True
            return
        jitter_extra, tries = 0.0, 0
        while True:
            try:
                if self._eval_hook is not None:
                    # (the engine's own factorisation is NOT at these parameters afterwards: whoever installs the hook marks the
                    # model dirty when taking it out)
                    res = self._eval_hook(theta, noise, CONST_JITTER + jitter_extra)
                    res = res if want_grad else res[0]
                else:
                    res = self._engine.eval(theta, noise, CONST_JITTER + jitter_extra, want_grad=want_grad)
                self._count_eval()
                self._main_evals += 1
                break
            except NotPositiveDefinite:
                self._count_eval()
                self._count_retry()
                self._main_evals += 1
                tries += 1
                diag_mean = self.kern.Kdiag_value() + noise + CONST_JITTER
                if tries > 5 or not np.isfinite(diag_mean):
                    raise np.linalg.LinAlgError("not positive definite, even with jitter.")
                jitter_extra = diag_mean * 1e-6 * 10 ** (tries - 1)
        if want_grad:
            self._nlml, self._grad_nat = res
        else:
            self._nlml, self._grad_nat = res, None
        self._have_grad = want_grad
        self._jitter_used = CONST_JITTER + jitter_extra
        self._dirty = False

    def _count_eval(self):
        with self._evals_lock:       # (+= on an attribute is a read and a write: two lanes' threads would lose counts now and then)
            self.n_evals += 1

    def _count_retry(self):
        with self._evals_lock:       # a jitter retry (GPy's jitchol: Ky not positive definite, the evaluation repeated with more jitter)
            self.n_jitter_retries += 1

    def objective_function(self):
        self._ensure(False)
        return self._nlml

    def log_likelihood(self):
        return -self.objective_function()

    def _natural_gradients(self):
        """dNLML/d(param) for every distinct Param (shared params accumulate)"""
        self._ensure(True)
        g = self._grad_nat
        acc, k = {}, 0
        for v, ls in self._part_params:      # gradient layout = theta layout (variance, lengthscale(s) per part), noise last
            for p in [v] + ls:
                acc[id(p)] = acc.get(id(p), 0.0) + g[k]
                k += 1
        acc[id(self.likelihood.variance)] = g[-1]
        for p in self.parameters():
            p.gradient = -acc[id(p)]  # GPy stores d log-likelihood / d param
        return acc

    def objective_function_gradients(self):
        acc = self._natural_gradients()
        free = self._free_params()
        g = np.array([acc[id(p)] for p in free])
        return _logexp_gradfactor(np.array([p.value for p in free]), g)

    def _objective_grads(self, x):
        try:
            self.optimizer_array = x
            self._ensure(True)  # ONE fused GPU evaluation: objective and gradient together
            f = self.objective_function()
            g = self.objective_function_gradients()
            self._fail_count = 0
            self._last_good_grad = g
        except (np.linalg.LinAlgError, ZeroDivisionError, ValueError):
            if self._fail_count >= self._allowed_failures:
                raise
            self._fail_count += 1
            with self._evals_lock:
                self.n_failed_evals += 1
            stale = getattr(self, "_last_good_grad", None)
            if stale is None or len(stale) != len(x):
                stale = np.zeros_like(x)
            return _F_FAILED, np.clip(stale, -_G_CLIP_FAILED, _G_CLIP_FAILED)
        return f, np.clip(g, -1e100, 1e100)

    # ---- optimisation --------------------------------------------------------------------------------
    def optimize(self, optimizer=None, max_iters=1000, messages=False, **kwargs):
        """L-BFGS-B on the softplus-transformed free parameters (paramz opt_lbfgsb [GPy-recall]:
        fmin_l_bfgs_b(f_fp, x0, maxfun=max_iters, maxiter=max_iters), scipy defaults m=10, factr=1e7, pgtol=1e-5)."""
        if optimizer is not None and "bfgs" not in str(optimizer).lower() and str(optimizer).lower() != "scg":
            raise NotImplementedError("only the (L-)BFGS(-B) optimiser of the reference recipe is provided")
        x0 = self.optimizer_array.copy()
        if x0.size == 0:
            return None
        if _lbfgsb.available() and not messages and self._eval_hook is None and self.fast_optimize:
            # The run as a generator of engine evaluation requests (optimize_program: scipy's L-BFGS-B core by reverse communication,
            # the objective on index tables instead of Param objects), every request evaluated on the spot.  The same steps as the
            # path below, bit for bit (tests/test_reference_l3.py replays the recorded evaluations of whole fits through it); per
            # evaluation ~50 us less Python -- a fit at the reference's own sizes is thousands of evaluations of ~55 us of GPU time.
            got = []
            prog = self.optimize_program(max_iters, got)
            try:
                req = next(prog)
                while True:
                    try:
                        res = self._engine.eval(req[0], req[1], req[2], want_grad=True)
                    except _EVAL_ERRORS as ex:       # a failed Cholesky (jitter retry) or any other failed evaluation: the objective's call
                        req = prog.throw(ex)
                    else:
                        req = prog.send(res)
            except StopIteration:
                pass
            return got[0]
        m0 = self._main_evals   # this run's own evaluations (self.n_evals also counts background restarts running beside it)
        fun, budget = _capped(self._objective_grads, self.eval_cap, x0)
        try:
            if _lbfgsb.available() and not messages:
                x_opt, f_opt, d = _lbfgsb.minimize(fun, x0, maxfun=int(max_iters), maxiter=int(max_iters))
            else:
                x_opt, f_opt, d = _sciopt.fmin_l_bfgs_b(fun, x0, maxfun=int(max_iters), maxiter=int(max_iters),
                                                        iprint=1 if messages else -1)
        except _BudgetExhausted:
            x_opt, f_opt, d = budget["x"], budget["f"], {"task": "STOP: evaluation cap reached"}
        self.optimizer_array = x_opt
        run = _OptRun(np.array(x_opt), float(f_opt), self._main_evals - m0, d.get("task", d.get("warnflag")))
        self.optimization_runs.append(run)
        return run

    def _stateless_objective_gen(self, free, counter=None, carry=False):
        """-> f_gen(x): the NLML and its optimizer-space gradient at x as a GENERATOR that does not evaluate anything itself: it yields
        the engine evaluations it needs -- (theta, noise, jitter) -- and is sent their results (nlml, grad), or has the engine's
        NotPositiveDefinite thrown in (GPy's jitter retries: up to five more requests); its return value is (f, g).  Nothing of the
        model's Param objects is touched (free = the parameters x stands for; every other parameter keeps the value it has NOW).
        Whoever drives the generator decides how the requests are evaluated: one by one on an engine handle (`_stateless_objective`)
        or, for several runs in lock step, in one batched pass per round (`LockstepLane`).  `counter`: a one-element list that
        counts the engine evaluations asked for.  `carry`: the failed-evaluation state -- consecutive failures and the last good
        gradient -- lives ON THE MODEL (`_fail_count`, `_last_good_grad`), as paramz keeps it, so that it persists across the
        model's own SEQUENTIAL runs (optimize(), first run -> restart 0, the sequential order of the restarts); a randomized restart
        that runs beside others (lock step, background handles) keeps the state per run -- its order among the others is not
        defined, a shared count would not be either."""
        # index tables built once: the per-evaluation work is a handful of small-array operations (it runs under the GIL, beside
        # the other runs of a lock-stepped fit: every microsecond here is GPU idle time times the number of runs)
        params = self.parameters()
        pos = {id(p): k for k, p in enumerate(params)}
        base = np.array([p.value for p in params], dtype=np.float64)
        free_idx = np.array([pos[id(p)] for p in free], dtype=np.intp)
        theta_idx = np.array([pos[id(q)] for v, ls in self._part_params for q in [v] + ls], dtype=np.intp)
        term_idx = [np.array([pos[id(f.variance)] for f in term], dtype=np.intp) for term in self.kern._terms()]
        noise_idx = pos[id(self.likelihood.variance)]
        shared = len(set(theta_idx.tolist())) != len(theta_idx)      # a Param object used by several factors: gradients add up
        state = {"fails": 0, "g": None}
        if carry:
            stale0 = getattr(self, "_last_good_grad", None)
            state = {"fails": int(self._fail_count), "g": stale0 if stale0 is not None and len(stale0) == len(free) else None}

        def f_gen(x):
            vals = base.copy()
            pv = _logexp_f(np.asarray(x, dtype=np.float64))
            vals[free_idx] = pv
            theta = vals[theta_idx]
            noise = float(vals[noise_idx])
            jitter_extra, tries = 0.0, 0
            try:
                _check_parameters(theta, noise)
                while True:
                    try:
                        nlml, g = yield (theta, noise, CONST_JITTER + jitter_extra)
                        self._count_eval()
                        if counter is not None:
                            counter[0] += 1
                        break
                    except NotPositiveDefinite:
                        self._count_eval()
                        self._count_retry()
                        if counter is not None:
                            counter[0] += 1
                        tries += 1
                        diag_mean = sum(float(np.prod(vals[t])) for t in term_idx) + noise + CONST_JITTER
                        if tries > 5 or not np.isfinite(diag_mean):
                            raise np.linalg.LinAlgError("not positive definite, even with jitter.")
                        jitter_extra = diag_mean * 1e-6 * 10 ** (tries - 1)
                state["fails"] = 0
                if carry:
                    self._fail_count = 0
            except (np.linalg.LinAlgError, ZeroDivisionError, ValueError):
                if state["fails"] >= self._allowed_failures:
                    raise
                state["fails"] += 1
                if carry:
                    self._fail_count = state["fails"]
                with self._evals_lock:
                    self.n_failed_evals += 1
                stale = state["g"] if state["g"] is not None else np.zeros_like(pv)
                return _F_FAILED, np.clip(stale, -_G_CLIP_FAILED, _G_CLIP_FAILED)
            acc = np.zeros(len(params))
            if shared:
                np.add.at(acc, theta_idx, g[:-1])
            else:
                acc[theta_idx] = g[:-1]
            acc[noise_idx] = g[-1]
            gf = _logexp_gradfactor(pv, acc[free_idx])
            state["g"] = gf
            if carry:
                self._last_good_grad = gf
            return nlml, np.clip(gf, -1e100, 1e100)

        return f_gen

    # ---- stateless objective: lets independent L-BFGS-B runs proceed concurrently on separate engines --------
    def _stateless_objective(self, eng, free, evaluate=None):
        """-> f_fp(x): `_stateless_objective_gen` driven on the spot -- every request evaluated by `eng.eval`, or by
        `evaluate(theta, noise, jitter) -> (nlml, grad)` where given (a LockstepEvaluator slot)."""
        f_gen = self._stateless_objective_gen(free)

        def f_fp(x):
            gen = f_gen(x)
            try:
                req = next(gen)
                while True:
                    try:
                        res = evaluate(*req) if evaluate is not None else eng.eval(req[0], req[1], req[2], want_grad=True)
                    except _EVAL_ERRORS as ex:
                        req = gen.throw(ex)
                    else:
                        req = gen.send(res)
            except StopIteration as stop:
                return stop.value

        return f_fp

    def _run_gen(self, free, x0, max_iters, counter=None, carry=False):
        """one L-BFGS-B run (scipy's core by reverse communication: lbfgsb.Lbfgsb; controls, `eval_cap` and result as
        fmin_l_bfgs_b(f_fp, x0, maxfun = maxiter = max_iters) behind `_capped`) as a generator of engine evaluation requests;
        returns (x_opt, f_opt, task)"""
        f_gen = self._stateless_objective_gen(free, counter, carry)
        run = _lbfgsb.Lbfgsb(x0, maxfun=int(max_iters), maxiter=int(max_iters))
        cap, n = self.eval_cap, 0
        best_f, best_x = np.inf, np.array(x0, dtype=np.float64)
        while True:
            x = run.ask()
            if x is None:
                break
            if cap:
                if n >= cap:
                    run.stop(best_x, best_f)
                    break
                n += 1
            f, g = yield from f_gen(x)
            if cap and f < best_f:
                best_f, best_x = float(f), np.array(x, dtype=np.float64)
            run.tell(f, g)
        return np.array(run.x), float(run.f), run.message

    def start_background_restarts(self, indices, engines, free=None, rand_gen=None, max_iters=1000, spare=0):
        """Run the randomized restarts `indices` (each: N(0,1) start in optimizer space, L-BFGS-B with
        maxfun = maxiter = max_iters) on the given auxiliary engines, one thread per engine; returns a handle
        whose .result() gives [(f_opt, x_opt, index), ...].  `free` = the parameters that are free DURING the
        restarts (default: all).  The restarts of the reference recipe do not depend on the run that precedes
        them (paramz randomizes every free parameter), so they may overlap it; with two or three evaluations
        in flight the GPU's idle phases (the serial Cholesky chain) of one are filled by the bulk work of another.
        `spare`: worker threads beyond the engines given now -- for engines lent later (`handle.lend(engine)`: the caller's own
        main engine once its sequential runs are through, so that the tail of the restarts does not run alone)."""
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor
        free = list(free) if free is not None else self.parameters()
        indices = list(indices)
        if not indices or not engines:
            class _Done:
                def result(self_inner):
                    return []
            return _Done()
        for e in engines:
            e.set_data(self.X, self.Y[:, 0])
            e.set_kernel(self._parts)
        starts = {}
        for i in indices:  # draw in index order (deterministic with a seeded rand_gen(i))
            gen = rand_gen(i) if callable(rand_gen) else None
            draw = gen(size=len(free)) if gen is not None else np.random.normal(size=len(free))
            # same round trip through the parameter domain as randomize() + optimize(): bit-identical start
            starts[i] = _logexp_finv(_logexp_f(draw))
        pool_q = queue.Queue()
        for e in engines:
            pool_q.put(e)
        lock = threading.Lock()

        def one(i):
            eng = pool_q.get()
            try:
                f_fp, budget = _capped(self._stateless_objective(eng, free), self.eval_cap, starts[i])
                try:
                    x_opt, f_opt, _ = _sciopt.fmin_l_bfgs_b(f_fp, starts[i], maxfun=int(max_iters), maxiter=int(max_iters))
                except _BudgetExhausted:
                    x_opt, f_opt = budget["x"], budget["f"]
                with lock:
                    self.optimization_runs.append(_OptRun(np.array(x_opt), float(f_opt), -1, "background", background=True))
                return float(f_opt), np.array(x_opt), i
            finally:
                pool_q.put(eng)

        ex = ThreadPoolExecutor(max_workers=len(engines) + max(0, int(spare)))
        futs = [ex.submit(one, i) for i in indices]
        model = self

        class _Handle:
            def lend(self_inner, eng):
                """one more engine for the restarts still waiting (its data / kernel are set here; whatever state it held is gone)"""
                if all(f.done() for f in futs):
                    return False
                eng.set_data(model.X, model.Y[:, 0])
                eng.set_kernel(model._parts)
                pool_q.put(eng)
                return True

            def result(self_inner):
                try:
                    return [f.result() for f in futs]
                finally:
                    ex.shutdown(wait=True)
        return _Handle()

    def restart_starts(self, indices, free, rand_gen=None):
        """-> {index: start of randomized restart `index` in optimizer space}: the N(0,1) draw of paramz' randomize() taken through
        the parameter domain and back, as randomize() + optimize() do (bit-identical start); drawn in index order (deterministic
        with a seeded rand_gen(i))"""
        starts = {}
        for i in indices:
            gen = rand_gen(i) if callable(rand_gen) else None
            draw = gen(size=len(free)) if gen is not None else np.random.normal(size=len(free))
            starts[i] = _logexp_finv(_logexp_f(draw))
        return starts

    def restart_program(self, take_index, starts, free, max_iters, out):
        """one slot of a lock-stepped fit as a generator of engine evaluation requests (LockstepLane.drive): it takes the next
        randomized restart still waiting (`take_index() -> index or None`), runs it (`_run_gen`), records it, and goes on until
        none is left.  out[index] = (f_opt, x_opt, index)."""
        while True:
            i = take_index()
            if i is None:
                return
            counter = [0]
            x_opt, f_opt, task = yield from self._run_gen(free, starts[i], max_iters, counter)
            self.optimization_runs.append(_OptRun(x_opt, f_opt, counter[0], task, background=True))
            out[i] = (f_opt, x_opt, i)

    def optimize_program(self, max_iters, result=None):
        """`optimize(max_iters=...)` of THIS model as a generator of engine evaluation requests: the run starts at the model's
        current point over its currently free parameters and installs its optimum in the model at the end, as optimize() does --
        but its evaluations are requests to whoever drives the generator (a LockstepLane), through the stateless objective (the
        same arithmetic as the model's own, bit for bit: it is what the randomized restarts run on).  The engine handle's own
        factorisation is not at any of the points evaluated: the model is left dirty.  result: a list that receives the _OptRun."""
        x0 = self.optimizer_array.copy()
        if x0.size == 0:
            return
        counter = [0]
        x_opt, f_opt, task = yield from self._run_gen(self._free_params(), x0, max_iters, counter, carry=True)
        self.optimizer_array = x_opt
        self._main_evals += counter[0]
        run = _OptRun(x_opt, f_opt, counter[0], task)
        self.optimization_runs.append(run)
        self._dirty = True
        self._have_grad = False
        if result is not None:
            result.append(run)

    def start_lockstep_restarts(self, indices, lanes, free=None, rand_gen=None, max_iters=1000):
        """The randomized restarts `indices` as lock-stepped runs.  `lanes` = [(LockstepEvaluator, [slot, ...]), ...]: one thread
        per slot, every evaluation through its lane's `evaluate(slot, ...)` (a lane = one engine handle with this model's data
        and its own rounds; the caller's own sequential runs may hold a slot of a lane too); a slot takes the next restart
        still waiting when its run ends and retires when none is left.  Same draws, same L-BFGS-B controls, same `eval_cap`
        as start_background_restarts; .result() -> [(f_opt, x_opt, index), ...] after every slot has retired."""
        import queue
        import threading
        free = list(free) if free is not None else self.parameters()
        indices = list(indices)
        starts = {}
        for i in indices:  # draw in index order (deterministic with a seeded rand_gen(i))
            gen = rand_gen(i) if callable(rand_gen) else None
            draw = gen(size=len(free)) if gen is not None else np.random.normal(size=len(free))
            starts[i] = _logexp_finv(_logexp_f(draw))   # the round trip randomize() + optimize() makes: bit-identical start
        lock = threading.Lock()
        out, errors = {}, {}
        todo = queue.Queue()
        for i in indices:
            todo.put(i)

        def worker(lockstep, slot):
            try:
                while True:
                    try:
                        i = todo.get_nowait()
                    except queue.Empty:
                        return
                    try:
                        one(i, lockstep, slot)
                    except BaseException as ex:  # noqa: BLE001 - reported by result(); the slot goes on to the next restart
                        errors[i] = ex
            finally:
                lockstep.retire(slot)

        def one(i, lockstep, slot):
            count = {"n": 0}

            def evaluate(theta, noise, jitter):
                count["n"] += 1
                return lockstep.evaluate(slot, theta, noise, jitter)

            f_fp, budget = _capped(self._stateless_objective(None, free, evaluate=evaluate), self.eval_cap, starts[i])
            try:
                x_opt, f_opt, d = _sciopt.fmin_l_bfgs_b(f_fp, starts[i], maxfun=int(max_iters), maxiter=int(max_iters))
                task = d.get("task", d.get("warnflag"))
            except _BudgetExhausted:
                x_opt, f_opt, task = budget["x"], budget["f"], "STOP: evaluation cap reached"
            with lock:
                self.optimization_runs.append(_OptRun(np.array(x_opt), float(f_opt), count["n"], task, background=True))
                out[i] = (float(f_opt), np.array(x_opt), i)

        threads = [threading.Thread(target=worker, args=(ls, slot), daemon=True) for ls, slots in lanes for slot in slots]
        for t in threads:
            t.start()

        class _Handle:
            def result(self_inner):
                for t in threads:
                    t.join()
                if errors:
                    raise errors[min(errors)]
                return [out[i] for i in indices]
        return _Handle()

    def lend_engine(self, handle):
        """lend this model's OWN engine to its background restarts (after the model's sequential runs): the factorisation it
        holds is given up -- the next use of the model refactorises at the then-current parameters."""
        if handle is None or not hasattr(handle, "lend"):
            return False
        if handle.lend(self._engine):
            self._dirty = True
            self._have_grad = False
            return True
        return False

    def optimize_restarts(self, num_restarts=10, robust=False, verbose=True, parallel=False, num_processes=None,
                          rand_gen=None, comm=None, **kwargs):
        """Restart 0 continues from the current point; restarts 1.. randomize first; the best f_opt wins
        (paramz Model.optimize_restarts [GPy-recall]; call site src/abstractMFGP.py:137).

        comm: optional sharding.Comm -- the restarts are independent L-BFGS-B runs on the same (X, Y),
        so rank r runs restarts r, r+size, ... and the (f_opt, x_opt) pairs are all-gathered.
        With a seeded per-restart `rand_gen(i)` the winner does not depend on the number of ranks."""
        initial_parameters = self.optimizer_array.copy()
        rank, size = (comm.rank, comm.size) if comm is not None else (0, 1)
        mine = []
        for i in range(num_restarts):
            if i % size != rank:
                continue
            try:
                if i > 0:   # restart 0 IS the current point (no reset: a round trip through the transform costs the last bits)
                    if callable(rand_gen):
                        self.randomize(rand_gen(i))
                    else:
                        self.randomize(None)
                run = self.optimize(**kwargs)
                if run is not None:
                    mine.append((run.f_opt, run.x_opt, i))
                if verbose:
                    print("Optimization restart %d/%d, f = %s" % (i + 1, num_restarts, run.f_opt if run else None))
            except Exception as e:  # noqa: BLE001 - mirrors paramz' robust mode
                if robust:
                    print("Warning - optimization restart %d/%d failed: %s" % (i + 1, num_restarts, e))
                else:
                    raise
        runs = mine
        if comm is not None and size > 1:
            runs = [r for part in comm.allgather_object(mine) for r in part]
        if runs:
            best = min(runs, key=lambda r: (r[0], r[2]))
            self.optimizer_array = best[1]
        else:
            self.optimizer_array = initial_parameters

    # ---- incremental data ------------------------------------------------------------------------------
    def append(self, x_new, y_new):
        """add ONE training row at the current hyper-parameters: O(N^2) rank-1 extension of the
        factorisation on the device when a padding slot is free, full re-upload + lazy refactorisation otherwise
        (and whenever the parameters changed since the last factorisation).  The numbers equal a fresh
        factorisation of the extended data at the same hyper-parameters."""
        x_new = np.asarray(x_new, dtype=np.float64).reshape(1, -1)
        y_new = float(np.asarray(y_new).reshape(-1)[0])
        done = False
        if not self._dirty:
            try:
                done = self._engine.append_row(x_new[0], y_new)
            except NotPositiveDefinite:
                done = False
        self.X = np.vstack([self.X, x_new])
        self.Y = np.vstack([self.Y, [[y_new]]])
        if done:
            self._nlml = self._engine.nlml()
            self._have_grad = False
        else:
            self._engine.set_data(self.X, self.Y[:, 0])
            self._dirty = True
        return done

    # ---- prediction ------------------------------------------------------------------------------------
    def predict(self, Xnew, full_cov=False, Y_metadata=None, kern=None, likelihood=None, include_likelihood=True):
        """-> (mean (N*,1), variance (N*,1)); the variance includes the noise variance and its latent
        part is floored at 1e-15 (GPy Posterior._raw_predict + Gaussian.predictive_values [GPy-recall])."""
        if full_cov:
            raise NotImplementedError("full_cov is not used by the reference")
        Xnew = np.ascontiguousarray(Xnew, dtype=np.float64)
        self._ensure(False)
        mean, var = self._engine.predict(Xnew, want_var=True, include_noise=include_likelihood)
        return mean[:, None], var[:, None]

    def predict_mean(self, Xnew):
        """mean only (what `lambda t: lf_model.predict(t)[0]` needs, src/abstractMFGP.py:104): skips the O(N^2 N*) variance"""
        Xnew = np.ascontiguousarray(Xnew, dtype=np.float64)
        self._ensure(False)
        mean, _ = self._engine.predict(Xnew, want_var=False)
        return mean[:, None]

    def augment(self, X, offsets):
        """[X | posterior mean of THIS level at X + offsets[j]]: the next level's inputs in one device call
        (src/MFDataFusion.py:177-208 with f_low = lambda t: lf_model.predict(t)[0], src/abstractMFGP.py:104)."""
        self._ensure(False)
        return self._engine.augment(X, offsets)

    def predict_chained(self, lf_model, Xnew, offsets, include_likelihood=True):
        """predict() of this level at the rows augmented by lf_model's posterior mean, the hand-over kept on the device
         -> (mean (N*,1), variance (N*,1)); same numbers as predict(lf_model.augment(Xnew, offsets))."""
        lf_model._ensure(False)
        self._ensure(False)
        mean, var = self._engine.predict_chained(lf_model._engine, Xnew, offsets, want_var=True,
                                                 include_noise=include_likelihood)
        return mean[:, None], var[:, None]

    def timings(self):
        return self._engine.timings()

    def close(self):
        if self._owns_engine:
            self._engine.close()

    def __str__(self):
        rows = ["%-40s %.6g%s" % (n, p.value, "  (fixed)" if p.fixed else "") for n, p in self._named_parameters()]
        return "Name : %s\nObjective : %s\n" % (self.name, self._nlml) + "\n".join(rows)
