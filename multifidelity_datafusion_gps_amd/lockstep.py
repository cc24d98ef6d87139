"""Lock-stepped L-BFGS-B runs over batched evaluations (mfgp_eval_batch): the objective's failure policy constants, the run record,
LockstepEvaluator (a thread per run on scipy's public function) and LockstepLane (one loop over run generators: scipy's L-BFGS-B
core by reverse communication, lbfgsb.py).  The reference's recipe is 1 + 6 runs on the same data
(/root/reference/src/abstractMFGP.py:131-137); here the live runs of a lane share one pass of the plan per step -- same runs, same
steps, bit for bit.  Split out of engine.py in round 6; engine.py re-exports every name."""
import time

import numpy as np

from . import _lib
from ._lib import NotPositiveDefinite



# ------------------------------------------------------------------------------------------------
# the model
# ------------------------------------------------------------------------------------------------
class _BudgetExhausted(Exception):
    pass


_F_FAILED = np.finfo(np.float64).max   # objective reported for a failed evaluation [GPy-recall: paramz Model._objective_grads
_G_CLIP_FAILED = 1e10                  # returns DBL_MAX, not inf -- an infinite value turns L-BFGS-B's cubic line-search
                                       # interpolation into NaN steps -- and the previous gradient clipped to +-1e10]


# what a failed evaluation raises (paramz Model._objective_grads catches exactly these [GPy-recall]): thrown INTO the objective generator
# by whoever drives it, so that the objective's own policy (jitter retries, DBL_MAX and the previous gradient) deals with them
_EVAL_ERRORS = (np.linalg.LinAlgError, ZeroDivisionError, ValueError)


def _check_parameters(theta, noise):
    """a NaN / infinite / non-positive parameter (a line search gone astray) is a FAILED evaluation, handled like a failed
    Cholesky (GPy: the NaNs end in jitchol's LinAlgError), not an argument error of the engine"""
    if not (np.all(np.isfinite(theta)) and np.all(theta > 0.0) and np.isfinite(noise) and noise >= 0.0):
        raise np.linalg.LinAlgError("hyper-parameters left the positive finite domain")


def _capped(f_fp, cap, x0):
    """-> (f, state): f evaluates f_fp at most `cap` times and then raises _BudgetExhausted; state holds the best point
    seen.  scipy's maxfun is only checked between iterations, so a run may overshoot it by a line search; a benchmark
    that compares code versions at a FIXED evaluation budget needs the count exact (cap = None: no cap)."""
    state = {"n": 0, "f": np.inf, "x": np.array(x0, dtype=np.float64)}
    if not cap:
        return f_fp, state

    def f(x):
        if state["n"] >= cap:
            raise _BudgetExhausted()
        state["n"] += 1
        val, g = f_fp(x)
        if val < state["f"]:
            state["f"], state["x"] = float(val), np.array(x, dtype=np.float64)
        return val, g
    return f, state


class _OptRun:
    def __init__(self, x_opt, f_opt, n_evals, status, background=False):
        self.x_opt, self.f_opt, self.n_evals, self.status = x_opt, f_opt, n_evals, status
        self.background = background     # a randomized restart that ran beside the model's own sequential runs


class LockstepEvaluator:
    """Independent L-BFGS-B runs on ONE engine handle, one evaluation per run and round: every run asks for its next objective
    (+ gradient) through `evaluate` and blocks; when all runs still alive have asked, the round goes to the GPU as ONE batched
    pass (`Engine.eval_batch`: the B matrix sets side by side in every launch of the factorisation sweep) and everybody gets
    its own result back.  The restarts of the reference's recipe (optimize_restarts(6, ...), src/abstractMFGP.py:137) are such
    runs: by paramz' semantics they start from fresh N(0,1) draws and never look at each other.  A batched evaluation is
    bitwise the single one, so every run takes exactly the steps it takes alone -- only the wall clock changes: at N <= 4096 one
    evaluation leaves most of the GPU idle (its serial Cholesky chain), B of them cost little more than one."""

    def __init__(self, engine, n_slots):
        import threading
        self._eng = engine
        self._cv = threading.Condition()
        self._active = int(n_slots)
        self._pending = {}
        self._results = {}
        self.rounds = 0
        self.evals = 0
        self.round_sizes = []
        self.engine_s = 0.0       # wall seconds inside eval_batch (the rest of a fit's time is the hosts' L-BFGS-B steps and hand-offs)
        self.oom_fallbacks = 0    # rounds whose batch did not fit the device and went request by request

    def evaluate(self, slot, theta, noise, jitter):
        """-> (nlml, grad) of THIS slot's point; raises NotPositiveDefinite for it alone"""
        with self._cv:
            self._pending[slot] = (np.array(theta, dtype=np.float64), float(noise), float(jitter))
            if len(self._pending) >= self._active:
                self._run_round()
            while slot not in self._results:
                self._cv.wait()
            res = self._results.pop(slot)
        if isinstance(res, BaseException):
            raise res
        return res

    def retire(self, slot):
        """this slot's run is over (it asks for nothing more): the others no longer wait for it"""
        with self._cv:
            self._active -= 1
            if self._pending and len(self._pending) >= self._active:
                self._run_round()

    def _run_round(self):
        # called with the lock held; every live run is blocked in evaluate(), so nothing else touches the engine
        slots = sorted(self._pending)
        reqs = [self._pending.pop(k) for k in slots]
        cap = getattr(self._eng, "MAX_BATCH", 16)
        t0 = time.perf_counter()
        try:
            for c0 in range(0, len(slots), cap):
                part, sl = reqs[c0:c0 + cap], slots[c0:c0 + cap]
                try:
                    nlml, grads, status = self._eng.eval_batch(np.array([r[0] for r in part]), [r[1] for r in part],
                                                               [r[2] for r in part], want_grad=True)
                except _lib.EngineOutOfMemory:       # the sets do not fit: request by request on the handle's own slab (same results)
                    self.oom_fallbacks += 1
                    for r, k in zip(part, sl):
                        try:
                            f, g = self._eng.eval(r[0], r[1], r[2], want_grad=True)
                            self._results[k] = (float(f), np.array(g))
                        except NotPositiveDefinite as ex:
                            self._results[k] = ex
                    continue
                for j, k in enumerate(sl):
                    self._results[k] = (NotPositiveDefinite(int(status[j])) if status[j] != 0
                                        else (float(nlml[j]), np.array(grads[j])))
        except BaseException as ex:  # noqa: BLE001 - an engine error ends every run of the round, not just the caller's
            for k in slots:
                self._results.setdefault(k, ex)
        self.engine_s += time.perf_counter() - t0
        self.rounds += 1
        self.evals += len(slots)
        self.round_sizes.append(len(slots))
        self._cv.notify_all()


class LockstepLane:
    """Several independent L-BFGS-B runs on ONE engine handle, advanced in lock step by ONE loop: every live run is a generator
    (`GPRegression._run_gen` and the programs built from it) that yields the engine evaluation it needs next; a round collects the
    requests of all live runs, evaluates them as one batched pass (`Engine.eval_batch`) and sends every run its own result.  No
    thread per run, no hand-off per evaluation (the form of `LockstepEvaluator`, kept for a scipy whose L-BFGS-B core cannot be
    driven by reverse communication): the host side of a round is the runs' own L-BFGS-B steps and nothing else.  Same statistics
    as LockstepEvaluator (rounds, evals, round_sizes, engine_s)."""

    def __init__(self, engine, max_batch=None):
        self._eng = engine
        self.rounds = 0
        self.evals = 0
        self.round_sizes = []
        self.engine_s = 0.0
        # memory policy (round 5): the most sets one pass may carry -- the engine's limit, or less where the caller sized it from the
        # device's free memory (AbstractMFGP._ard_lockstep) -- halved whenever the engine answers EngineOutOfMemory; at 1 the lane
        # evaluates request by request with eval(), which needs no batch slab.  A batched evaluation is bitwise the single one, so
        # the runs take the same steps at every width.
        cap = int(getattr(engine, "MAX_BATCH", 16))
        self.max_batch = cap if not max_batch else max(1, min(cap, int(max_batch)))
        self.oom_fallbacks = []       # (sets asked for, sets per pass from then on)

    def _evaluate(self, part):
        """-> the results of the requests `part` (at most max_batch of them): (nlml, grad) or the exception of that request"""
        if len(part) > self.max_batch:                # (the lane narrowed since the caller cut its chunks)
            out, c0 = [], 0
            while c0 < len(part):
                n = self.max_batch
                out += self._evaluate(part[c0:c0 + n])
                c0 += n
            return out
        while len(part) > 1 and self.max_batch > 1:
            try:
                nlml, grads, status = self._eng.eval_batch(np.array([r[0] for r in part]), [r[1] for r in part],
                                                           [r[2] for r in part], want_grad=True)
            except _lib.EngineOutOfMemory:
                n_ = min(len(part), self.max_batch)
                new = (n_ + 1) // 2 if n_ > 2 else 1          # 6 -> 3 -> 2 -> 1
                self.oom_fallbacks.append((len(part), new))
                self.max_batch = new
                return self._evaluate(part)
            except _EVAL_ERRORS as ex:               # the pass as a whole failed: a failed evaluation of every run in it
                return [ex] * len(part)
            return [NotPositiveDefinite(int(status[j])) if status[j] != 0 else (float(nlml[j]), np.array(grads[j]))
                    for j in range(len(part))]
        out = []
        for theta, noise, jitter in part:            # one request, or a lane narrowed to one set: the handle's own evaluation
            try:
                if self.max_batch > 1 or not hasattr(self._eng, "eval"):
                    nlml, grads, status = self._eng.eval_batch(np.array([theta]), [noise], [jitter], want_grad=True)
                    out.append(NotPositiveDefinite(int(status[0])) if status[0] != 0 else (float(nlml[0]), np.array(grads[0])))
                else:
                    f, g = self._eng.eval(theta, noise, jitter, want_grad=True)
                    out.append((float(f), np.array(g)))
            except _lib.EngineOutOfMemory:           # (not even one set fits: single evaluations from here on)
                self.oom_fallbacks.append((1, 1))
                self.max_batch = 1
                try:
                    f, g = self._eng.eval(theta, noise, jitter, want_grad=True)
                    out.append((float(f), np.array(g)))
                except (NotPositiveDefinite,) + _EVAL_ERRORS as ex:
                    out.append(ex)
            except (NotPositiveDefinite,) + _EVAL_ERRORS as ex:
                out.append(ex)
        return out

    def drive(self, programs):
        """run the generators to their end; a program that raises ends every program of the lane (the exception propagates)"""
        live = {}
        for k, prog in enumerate(programs):
            try:
                live[k] = (prog, next(prog))
            except StopIteration:
                pass
        while live:
            slots = sorted(live)
            results = {}
            t0 = time.perf_counter()
            c0 = 0
            while c0 < len(slots):
                sl = slots[c0:c0 + self.max_batch]
                for k, res in zip(sl, self._evaluate([live[k][1] for k in sl])):
                    results[k] = res
                c0 += len(sl)
            self.engine_s += time.perf_counter() - t0
            self.rounds += 1
            self.evals += len(slots)
            self.round_sizes.append(len(slots))
            for k in slots:
                prog = live[k][0]
                res = results[k]
                try:
                    req = prog.throw(res) if isinstance(res, BaseException) else prog.send(res)
                    live[k] = (prog, req)
                except StopIteration:
                    del live[k]
