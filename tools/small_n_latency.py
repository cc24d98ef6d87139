"""Per-evaluation latency at the reference's own problem sizes (N = 10 .. 512): engine call alone vs the whole
host path (transform + ctypes + engine + gradient chain rule) that L-BFGS-B sees."""
import sys, os, time
import os as _os; _os.environ.setdefault("MFGP_TIMING", "1")   # start / end stamps of a call at every size (off by default below Np = 4096)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multifidelity_datafusion_gps_amd import engine as gp
from tests import cases

for N in (10, 50, 128, 200, 512):
    rng = np.random.default_rng(N)
    X = rng.uniform(size=(N, 2)); Y = cases.hf_2d(X)[:, None]
    Xa = np.hstack([X, cases.lf_2d(X)[:, None]])
    k = gp.RBF(1, active_dims=[2]) * gp.RBF(2, active_dims=[0, 1]) + gp.RBF(2, active_dims=[0, 1])
    m = gp.GPRegression(Xa, Y, kernel=k)
    x = m.optimizer_array.copy()
    m._objective_grads(x)
    reps = 300
    t0 = time.perf_counter()
    for i in range(reps):
        m._objective_grads(x + 1e-6 * (i % 7))
    host = (time.perf_counter() - t0) / reps
    eng = m._engine
    theta, noise = np.ones(6), 0.5
    eng.eval(theta, noise)
    t0 = time.perf_counter()
    for i in range(reps):
        eng.eval(theta + 1e-6 * (i % 7), noise)
    raw = (time.perf_counter() - t0) / reps
    print("N=%4d  objective+gradient through the host layer %.1f us, engine call alone %.1f us (GPU %.1f us)" % (
        N, host * 1e6, raw * 1e6, eng.timings()["total_ms"] * 1e3), flush=True)
    m.close()
