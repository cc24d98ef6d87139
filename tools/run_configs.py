"""Runs the five BASELINE.json configurations end to end on the GPU (synthetic data of SURVEY 8(d)) and prints one
line each: sizes, wall time, per-evaluation stage times.  `--quick` shrinks the large ones for smoke runs."""
import os as _os; _os.environ.setdefault("MFGP_HW_QUEUES", "2"); _os.environ.setdefault("MFGP_STAGE_TIMING", "1")   # per-stage stamps at every size (a handle records none below Np = 4096 by default)
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multifidelity_datafusion_gps_amd as mf
from multifidelity_datafusion_gps_amd._lib import Engine
from tests import cases

ap = argparse.ArgumentParser()
ap.add_argument("--quick", action="store_true")
ap.add_argument("--evals", type=int, default=20)
args = ap.parse_args()
E = args.evals


class BudgetNARGP(mf.NARGP):
    """fixed evaluation budget per L-BFGS-B run; a subclass because the data-driven LF level is fitted in the constructor"""
    lf_max_iters = first_run_max_iters = restart_max_iters = E
    eval_cap = E
    restart_lockstep = os.environ.get("MFGP_RESTART_LOCKSTEP", "1") != "0"      # the default: lock-stepped runs over batched evaluations
    restart_concurrency = int(os.environ.get("MFGP_RESTART_CONC", "2"))          # (MFGP_RESTART_LOCKSTEP=0: round 3's concurrent restarts)


def col(f):
    return lambda x: f(x)[:, None]


# cfg1: 2-fidelity NARGP, 1-D Forrester, N_lf = 50 / N_hf = 10 (plumbing)
X_lf = np.linspace(0, 1, 50)[:, None]
_w = mf.NARGP(1, col(cases.forrester_hf), None, lf_X=X_lf[::5], lf_Y=col(cases.forrester_lf)(X_lf[::5]), seed=1)   # untimed: first use loads
_w.first_run_max_iters = _w.restart_max_iters = 3                                                           # the code objects
_w.fit(np.random.default_rng(1).uniform(size=(5, 1))); _w.predict(X_lf); _w.close()
t0 = time.perf_counter()
m = mf.NARGP(1, col(cases.forrester_hf), None, lf_X=X_lf, lf_Y=col(cases.forrester_lf)(X_lf), seed=0)
m.fit(np.random.default_rng(0).uniform(size=(10, 1)))
Xt = np.linspace(0, 1, 200)[:, None]
mse = m.get_mse(Xt, col(cases.forrester_hf)(Xt))
print("cfg1 NARGP 1-D Forrester N_lf=50 N_hf=10 (full recipe 500/1000 iters): %.0f ms, test MSE %.3g" % ((time.perf_counter() - t0) * 1e3, mse))
m.close()

# cfg2: single-fidelity GP, 3-D, N = 4096, RBF: K build + Cholesky (+ the rest of one evaluation)
e = Engine(0)
rng = np.random.default_rng(1)
N = 1024 if args.quick else 4096
X = rng.uniform(size=(N, 3)); Y = cases.hf_3d(X)
e.set_data(X, Y); e.set_kernel(cases.single(cases.RBF, 3))
th = np.array([1.0, 0.3]); nz = 0.01 * Y.var()
e.eval(th, nz)
e.eval(th, nz); t = e.timings()
e.factorize(th, nz); e.factorize(th, nz); tf = e.timings()      # "K build + Cholesky only": no gradient, no K^-1
print("cfg2 single GP 3-D N=%d RBF: K-build %.3f ms (%.0f GB/s); factorisation alone (Cholesky + inverse, mfgp_factorize) %.3f ms, "
      "total %.3f ms; objective+gradient evaluation: sweep incl. K^-1 %.3f ms, total %.3f ms"
      % (N, t["kbuild_ms"], t["kbuild_bytes"] / t["kbuild_ms"] / 1e6, tf["cholinv_ms"], tf["total_ms"], t["cholinv_ms"], t["total_ms"]))
e.close()

# cfg3: 2-fidelity NARGP, 4-D, N_lf = 16384 / N_hf = 4096, data-driven LF
n_lf, n_hf = (2048, 1024) if args.quick else (16384, 4096)
rng = np.random.default_rng(2)
X_lf = rng.uniform(size=(n_lf, 4)); X_hf = rng.uniform(size=(n_hf, 4)); Xs = rng.uniform(size=(4096, 4))
t0 = time.perf_counter()
m = BudgetNARGP(4, col(cases.hf_4d), None, lf_X=X_lf, lf_Y=col(cases.lf_4d)(X_lf), seed=2)
t1 = time.perf_counter()
m.fit(X_hf)
t2 = time.perf_counter()
mean, var = m.predict(Xs)
t3 = time.perf_counter()
print("cfg3 NARGP 4-D N_lf=%d N_hf=%d (budget %d evals/run): LF fit %.0f ms (%d evals), HF fit %.0f ms (%d evals), predict 4096: %.0f ms, MSE %.3g"
      % (n_lf, n_hf, E, (t1 - t0) * 1e3, m.lf_model.n_evals, (t2 - t1) * 1e3, m.hf_model.n_evals, (t3 - t2) * 1e3,
         float(np.mean((mean - col(cases.hf_4d)(Xs)) ** 2))))
m.close()

# cfg4: 3-fidelity data fusion, 2-D, N per level: level 3's f_low is level 2's posterior mean (f_low is any callable)
n = 1024 if args.quick else 8192
rng = np.random.default_rng(3)
f3 = lambda x: (np.sin(10 * x[:, 0]) ** 2 + np.cos(10 * x[:, 1]))[:, None]
f2 = lambda x: 1.5 * f3(x) + 3
f1 = lambda x: f2(x) - 1.2 * (np.sin(0.1 * np.pi * x[:, :1]) + np.sin(0.1 * np.pi * x[:, 1:2]))
X1, X2, X3 = (rng.uniform(size=(n, 2)) for _ in range(3))
t0 = time.perf_counter()
lvl2 = BudgetNARGP(2, f2, None, lf_X=X1, lf_Y=f1(X1), seed=3, name="level2")
lvl2.fit(X2)
lvl3 = BudgetNARGP(2, f3, lambda x: lvl2.predict(x)[0], seed=4, name="level3")
lvl3.fit(X3)
Xs = rng.uniform(size=(2048, 2))
mean, var = lvl3.predict(Xs)
print("cfg4 3-level fusion 2-D N=%d per level (budget %d evals/run): %.0f ms total, MSE %.3g"
      % (n, E, (time.perf_counter() - t0) * 1e3, float(np.mean((mean - f3(Xs)) ** 2))))
lvl2.close(); lvl3.close()

# cfg5: NARGP + entropy-reduction adaptation, 4-D, N_hf growing; candidate panels through the batched DIRECT
n_lf, n0, steps = (1024, 128, 3) if args.quick else (16384, 512, 4)
rng = np.random.default_rng(4)
X_lf = rng.uniform(size=(n_lf, 4))
m = BudgetNARGP(4, col(cases.hf_4d), None, lf_X=X_lf, lf_Y=col(cases.lf_4d)(X_lf), lf_hf_adapt_ratio=0, seed=5,
                adapt_maximizer=mf.DIRECT1Maximizer())      # ratio 0: adapt only the HF level (the reference's LF adaptation is unreachable)
m.fit(rng.uniform(size=(n0, 4)))
t0 = time.perf_counter()
m.adapt(steps)
t1 = time.perf_counter()
m.adapt(steps, reoptimize=False)
t2 = time.perf_counter()
print("cfg5 adaptation 4-D N_lf=%d N_hf=%d->%d: %d steps with refit %.0f ms/step; %d steps with rank-1 append %.1f ms/step (DIRECT evals/step ~%d)"
      % (n_lf, n0, len(m.hf_X), steps, (t1 - t0) * 1e3 / steps, steps, (t2 - t1) * 1e3 / steps, m.adapt_maximizer.last_info["nf"]))
# the same in the candidate-panel form BASELINE.json words it ("predictive-variance panels"): one two-level panel of N* = 65536
# Sobol points per acquisition, at the start of the growth and at its end (N_hf = 8192)
n_star = 4096 if args.quick else 65536
m.adapt_maximizer = mf.PanelMaximizer(n_candidates=n_star, seed=1)
m.adapt(1, reoptimize=False)          # warm-up: draws the panel, sizes the device buffers
t0 = time.perf_counter()
m.adapt(steps, reoptimize=False)
t1 = time.perf_counter()
line = "cfg5 panel form N*=%d: N_hf=%d acquisition + rank-1 append %.1f ms/step" % (n_star, len(m.hf_X), (t1 - t0) * 1e3 / steps)
if not args.quick:
    m.eval_cap = m.lf_max_iters = m.first_run_max_iters = m.restart_max_iters = 2     # a token fit: the panel is what is timed
    m.num_restarts = 1
    m.fit(rng.uniform(size=(8192, 4)))
    m.adapt(1, reoptimize=False)
    t0 = time.perf_counter()
    m.adapt(steps, reoptimize=False)
    t1 = time.perf_counter()
    Np = 8192 + 128
    line += "; N_hf=%d %.1f ms/step (variance product %.1f TFLOP/s incl. everything else)" % (
        len(m.hf_X), (t1 - t0) * 1e3 / steps, float(Np) * Np * n_star / ((t1 - t0) / steps) / 1e12)
print(line)
m.close()
