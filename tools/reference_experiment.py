"""The reference's own experiment (its tests/utils.py `__main__` and tests/test_mfgp_adapt_{2,3,4}d.py), run on the HIP path:
for each fusion model (NARGP, GPDF, GPDFC) fit on `num_hf` random high-fidelity points, then `num_adapts` rounds of 5
variance-driven acquisitions, refreshing the polynomial-chaos mean / variance of the fused posterior mean after every
round, against the analytic moments of the test function.  chaospy is replaced by the package's own Gauss-Legendre
projection (LegendreGPC: the quadrature grid goes through ONE predictive panel per refresh); nothing is plotted.

    python tools/reference_experiment.py [dim=4] [num_adapts=5]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import multifidelity_datafusion_gps_amd as mf
from multifidelity_datafusion_gps_amd.gpc import LegendreGPC, MFGP_GPC

# the reference's test functions (tests/test_mfgp_adapt_2d.py / _3d.py / _4d.py: products of sines, plus a constant in 4-D)
A = {2: [2.2 * np.pi, np.pi], 3: [3.2 * np.pi, 2.1 * np.pi, 1.2 * np.pi], 4: [np.pi] * 4}
SHIFT = {2: 0.0, 3: 5.0, 4: 5.0}
LF_AMP = {2: 1.2, 3: 0.25, 4: 0.25}
LF_W = {2: [0.1, 0.1], 3: [0.1, 0.05, 0.15], 4: [0.1, 0.05, 0.15, 0.2]}


def make_functions(dim):
    a, c, w, amp = A[dim], SHIFT[dim], LF_W[dim], LF_AMP[dim]

    def hf(x):
        x = np.atleast_2d(x)
        return (np.prod([np.sin(x[:, k] * a[k]) for k in range(dim)], axis=0) + c)[:, None]

    def lf(x):
        x = np.atleast_2d(x)
        return hf(x) - amp * np.sum([np.sin(x[:, k] * np.pi * w[k]) for k in range(dim)], axis=0)[:, None]

    return hf, lf


def analytical_mean(a, constant=0.0):      # tests/utils.py:14-17
    return float(np.prod([(1 - np.cos(ai)) / ai for ai in a]) + constant)


def analytical_var(a):                     # tests/utils.py:20-27
    m = analytical_mean(a)
    t1 = np.prod([0.5 - np.sin(2 * ai) / (4 * ai) for ai in a])
    t3 = 2 * m * np.prod([(np.cos(ai) - 1) / ai for ai in a]) * ((-1) ** (len(a) - 1))
    return float(t1 + m ** 2 + t3)


def create_model(method, dim, hf, lf, seed):   # tests/utils.py:38-47
    kw = dict(add_noise=True, seed=seed, adapt_maximizer=mf.DIRECT1Maximizer())
    if method == "GPDF":
        return mf.GPDF(dim, 0.001, 2, hf, lf, **kw)
    if method == "NARGP":
        return mf.NARGP(dim, hf, lf, **kw)
    return mf.GPDFC(dim, 0.001, 2, hf, lf, **kw)


def main():
    dim = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    num_adapts = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    order = 10 if dim < 4 else 6                                   # tests/utils.py:88-92
    hf, lf = make_functions(dim)
    rng = np.random.default_rng(10)                                # (the reference seeds numpy's global generator with 10)
    X_hf = rng.uniform(size=(5, dim))                              # create_data: num_hf = 5, num_test = 100
    X_test = rng.uniform(size=(100, dim))
    m_true, v_true = analytical_mean(A[dim], SHIFT[dim]), analytical_var(A[dim])
    print("dim %d, PCE order %d, analytic mean %.6f variance %.6f" % (dim, order, m_true, v_true))
    for method in ("NARGP", "GPDF", "GPDFC"):
        t0 = time.perf_counter()
        model = create_model(method, dim, hf, lf, seed=3)
        model.fit(X_hf)
        pce = LegendreGPC(lambda x: model.predict(x)[0], np.zeros(dim), np.ones(dim), polynomial_order=order,
                          quadrature_order=order)
        drv = MFGP_GPC(model, pce, num_adapts, 5, X_test=X_test, Y_test=hf(X_test))
        drv.adapt()
        dt = time.perf_counter() - t0
        n_eval = model.hf_model.n_evals if hasattr(model.hf_model, "n_evals") else -1
        print("%-6s cost %s" % (method, drv.cost_history))
        print("       rel. error of the mean     %s" % " ".join("%.2e" % abs((m - m_true) / m_true) for m in drv.mean_history))
        print("       rel. error of the variance %s" % " ".join("%.2e" % abs((v - v_true) / v_true) for v in drv.var_history))
        print("       test MSE                   %s" % " ".join("%.2e" % e for e in drv.mse_history))
        print("       %.2f s wall (fit + %d rounds of 5 acquisitions with refit + %d moment refreshes over %d quadrature nodes)"
              % (dt, num_adapts, num_adapts + 1, pce.quad_weights.size))
        model.close()


if __name__ == "__main__":
    main()
