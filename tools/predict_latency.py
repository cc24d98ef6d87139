"""Latency of one predict call (mean + variance) against the number of test rows; MFGP_SKINNY=0 disables the skinny
variance product for N* <= 64 (then every batch is padded to a 128-row tile GEMM)."""
import sys, os, time
import os as _os; _os.environ.setdefault("MFGP_TIMING", "1")   # start / end stamps of a call at every size (off by default below Np = 4096)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multifidelity_datafusion_gps_amd._lib import Engine
from tests import cases
e = Engine(0)
for N in (512, 2048, 8192):
    rng = np.random.default_rng(N)
    X = rng.uniform(size=(N, 4)); Y = cases.hf_4d(X)
    Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
    e.set_data(Xa, Y); e.set_kernel(cases.composite(4, 1))
    e.factorize(np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.01)
    for ns in (1, 16, 32, 64, 65, 128, 1000):
        Xs = Xa[:ns] + 0.01
        e.predict(Xs)
        t0 = time.perf_counter()
        for _ in range(20):
            e.predict(Xs)
        dt = (time.perf_counter() - t0) / 20
        t = e.timings()
        print("N=%5d N*=%5d  %.3f ms per predict call (panel %.3f var %.3f)" % (N, ns, dt * 1e3, t["predict_panel_ms"], t["predict_var_ms"]), flush=True)
