"""L-BFGS-B by reverse communication: scipy's own core (`scipy.optimize._lbfgsb.setulb`, the C port of Zhu / Byrd / Lu / Nocedal's
code that `scipy.optimize.fmin_l_bfgs_b` drives) behind an ask / tell object instead of a callback.

Why: the reference's recipe (src/abstractMFGP.py:131-137) is 1 + 6 independent L-BFGS-B runs, and this package evaluates the
objectives of all live runs in ONE batched GPU pass per round (`mfgp_eval_batch`).  `fmin_l_bfgs_b` is a blocking call that owns its
thread, so lock step used to mean one thread per run and a condition-variable hand-off per evaluation -- 100-160 us of GIL traffic
per evaluation, more than the GPU's share of a batched evaluation below N ~ 2048.  With ask / tell one loop drives all runs of a lane.

`Lbfgsb` repeats `scipy.optimize._lbfgsb_py._minimize_lbfgsb` (scipy 1.15) statement by statement -- the workspace arrays, the
evaluation at x0 before the first `setulb` call (ScalarFunction's constructor), the cache that answers a second request at an
unchanged x without a new evaluation, the `maxiter` / `maxfun` checks at NEW_X, the status words -- so a run takes the same steps and
ends at the same point, bit for bit, as `fmin_l_bfgs_b(func, x0, maxfun=, maxiter=)`; `self_check()` verifies exactly that against
the public function once per process, and everything falls back to `fmin_l_bfgs_b` on threads when the private core is missing or
answers differently (another scipy).  Bounds are not supported (paramz' transformed parameters are unbounded).
"""
import numpy as np

try:
    from scipy.optimize import _lbfgsb as _core
except Exception:  # noqa: BLE001 - any failure means: use the public, blocking interface
    _core = None

_STATUS = {0: "START", 1: "NEW_X", 2: "RESTART", 3: "FG", 4: "CONVERGENCE", 5: "STOP", 6: "WARNING", 7: "ERROR", 8: "ABNORMAL"}
try:
    from scipy.optimize._lbfgsb_py import task_messages as _TASK
except Exception:  # noqa: BLE001
    _TASK = {}


class Lbfgsb:
    """one run.  x = run.ask() -> the point whose objective and gradient are wanted next (None: the run is over); run.tell(f, g).
    After the end: .x, .f, .g, .nfev, .nit, .warnflag, .message as fmin_l_bfgs_b reports them."""

    def __init__(self, x0, maxfun=15000, maxiter=15000, m=10, factr=1e7, pgtol=1e-5, maxls=20):
        x0 = np.asarray(x0).ravel()
        n, = x0.shape
        self._m, self._factr, self._pgtol, self._maxls = int(m), float(factr), float(pgtol), int(maxls)
        self._maxfun, self._maxiter = int(maxfun), int(maxiter)
        self._nbd = np.zeros(n, np.int32)
        self._low = np.zeros(n, np.float64)
        self._up = np.zeros(n, np.float64)
        self.x = np.array(x0, dtype=np.float64)
        self.f = np.array(0.0, dtype=np.int32)           # (as scipy initialises it: never read before the first evaluation)
        self.g = np.zeros((n,), dtype=np.int32)
        self._wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
        self._iwa = np.zeros(3 * n, dtype=np.int32)
        self._task = np.zeros(2, dtype=np.int32)
        self._ln_task = np.zeros(2, dtype=np.int32)
        self._lsave = np.zeros(4, dtype=np.int32)
        self._isave = np.zeros(44, dtype=np.int32)
        self._dsave = np.zeros(29, dtype=np.float64)
        self.nit = 0
        self.nfev = 0
        self.done = False
        self.warnflag = None
        self.message = None
        self._x_eval = None            # the point of the last evaluation (ScalarFunction.x): a request at the same point is answered from it
        self._f_eval = self._g_eval = None
        self._stage = 0                # 0: the evaluation at x0 is outstanding, 1: inside the setulb loop, 2: a loop evaluation is outstanding

    def ask(self):
        if self.done:
            return None
        if self._stage == 0:           # ScalarFunction.__init__: f and g at x0 before anything else
            return np.copy(self.x)
        if self._stage == 2:
            raise RuntimeError("tell() the outstanding evaluation first")
        while True:
            self.g = self.g.astype(np.float64)
            _core.setulb(self._m, self.x, self._low, self._up, self._nbd, self.f, self.g, self._factr, self._pgtol, self._wa,
                         self._iwa, self._task, self._lsave, self._isave, self._dsave, self._maxls, self._ln_task)
            t = self._task[0]
            if t == 3:
                if self._x_eval is not None and np.array_equal(self.x, self._x_eval):
                    self.f, self.g = self._f_eval, self._g_eval          # (fun_and_grad at an unchanged x: no evaluation)
                    continue
                self._stage = 2
                return np.copy(self.x)
            elif t == 1:
                self.nit += 1
                if self.nit >= self._maxiter:
                    self._task[0] = 5
                    self._task[1] = 504
                elif self.nfev > self._maxfun:
                    self._task[0] = 5
                    self._task[1] = 502
            else:
                break
        self._finish()
        return None

    def tell(self, f, g):
        if self._stage == 1 or self.done:
            raise RuntimeError("no evaluation outstanding")
        f = float(f)
        g = np.atleast_1d(np.asarray(g, dtype=np.float64))
        self.nfev += 1
        self._x_eval = np.copy(self.x)
        self._f_eval, self._g_eval = f, g
        if self._stage == 2:
            self.f, self.g = f, g
        self._stage = 1

    def stop(self, x, f, message="STOP: evaluation cap reached"):
        """end the run from outside at the point given (the caller's own budget)"""
        self.x, self.f = np.array(x, dtype=np.float64), float(f)
        self.done, self.warnflag, self.message = True, 1, message

    def _finish(self):
        t = self._task[0]
        if t == 4:
            self.warnflag = 0
        elif self.nfev > self._maxfun or self.nit >= self._maxiter:
            self.warnflag = 1
        else:
            self.warnflag = 2
        self.message = _STATUS.get(int(t), str(int(t))) + ": " + _TASK.get(int(self._task[1]), str(int(self._task[1])))
        self.done = True


def minimize(func, x0, maxfun=15000, maxiter=15000):
    """fmin_l_bfgs_b(func, x0, maxfun=maxfun, maxiter=maxiter) through the ask / tell object -> (x, f, d)"""
    run = Lbfgsb(x0, maxfun=maxfun, maxiter=maxiter)
    while True:
        x = run.ask()
        if x is None:
            break
        run.tell(*func(x))
    return run.x, run.f, {"grad": run.g, "task": run.message, "funcalls": run.nfev, "nit": run.nit, "warnflag": run.warnflag}


_checked = None


def available():
    """the private core is there AND drives a run exactly as the public function does (checked once per process)"""
    global _checked
    if _checked is None:
        _checked = _core is not None and self_check()
    return _checked


def self_check():
    from scipy.optimize import fmin_l_bfgs_b

    def make():
        seen = []

        def rosen(x):        # a line search with several trial points per iteration, more iterations than the budget below allows
            seen.append(np.array(x))
            a, b = x[:-1], x[1:]
            f = float(np.sum(100.0 * (b - a * a) ** 2 + (1.0 - a) ** 2))
            g = np.zeros_like(x)
            g[:-1] += -400.0 * a * (b - a * a) - 2.0 * (1.0 - a)
            g[1:] += 200.0 * (b - a * a)
            return f, g
        return rosen, seen
    try:
        for x0, budget in ((np.array([-1.2, 1.0, 0.7, -0.3]), 25), (np.array([0.5, 0.5]), 1000), (np.array([3.0, -2.0, 1.0]), 7)):
            fa, sa = make()
            fb, sb = make()
            xa, va, da = fmin_l_bfgs_b(fa, x0, maxfun=budget, maxiter=budget)
            xb, vb, db = minimize(fb, x0, maxfun=budget, maxiter=budget)
            same = (np.array_equal(xa, xb) and va == vb and da["funcalls"] == db["funcalls"] and da["nit"] == db["nit"]
                    and da["warnflag"] == db["warnflag"] and len(sa) == len(sb) and all(np.array_equal(p, q) for p, q in zip(sa, sb)))
            if not same:
                return False
        return True
    except Exception:  # noqa: BLE001 - a changed signature, a missing symbol: the public interface it is
        return False
