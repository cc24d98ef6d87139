"""Latency of one two-level predict (low-fidelity mean on the stencil -> augmented rows -> high-fidelity mean+variance):
host hand-over (two engine calls + numpy concatenation) against mfgp_predict_chained."""
import sys, os, time
import os as _os; _os.environ.setdefault("MFGP_TIMING", "1")   # start / end stamps of a call at every size (off by default below Np = 4096)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multifidelity_datafusion_gps_amd._lib import Engine
from tests import cases

lf, hf = Engine(0), Engine(0)
d = 4
offs = np.zeros((1, d))
for N in (512, 2048, 8192):
    rng = np.random.default_rng(N)
    X = rng.uniform(size=(N, d))
    lf.set_data(X, cases.lf_4d(X)); lf.set_kernel(cases.single(cases.RBF, d))
    lf.factorize(np.array([1.0, 0.5]), 1e-3)
    hf.set_data(lf.augment(X, offs), cases.hf_4d(X)); hf.set_kernel(cases.composite(d, 1))
    hf.factorize(np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.01)
    for ns in (1, 16, 128, 1000, 8192):
        Xs = rng.uniform(size=(ns, d))
        reps = 20 if ns < 8192 else 5

        def host():
            stack = (Xs[:, None, :] + offs[None, :, :]).reshape(-1, d)
            aug = np.hstack([Xs, lf.predict(stack, want_var=False)[0].reshape(ns, 1)])
            return hf.predict(aug)

        def chained():
            return hf.predict_chained(lf, Xs, offs)

        res = {}
        for name, fn in (("host", host), ("chained", chained)):
            fn()
            t0 = time.perf_counter()
            for _ in range(reps):
                out = fn()
            res[name] = ((time.perf_counter() - t0) / reps * 1e3, out)
        assert np.array_equal(res["host"][1][0], res["chained"][1][0]) and np.array_equal(res["host"][1][1], res["chained"][1][1])
        print("N=%5d N*=%5d  host hand-over %.3f ms   chained %.3f ms   (x%.2f)" % (N, ns, res["host"][0], res["chained"][0],
                                                                                 res["host"][0] / res["chained"][0]), flush=True)
