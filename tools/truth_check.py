"""One size, three evaluations of the same quantities: the HIP engine, the fp64 oracle (GPy's algorithm: explicit-inverse variance and
the triangular form) and the quad-precision checker (oracle/quad_truth.c) -- errors of the first two against the third, in units of
the stated tolerances.  The -m gpu suite does this up to N = 4096 (tests/test_gpu_truth.py); this tool is for the sizes whose
quad-precision Cholesky takes minutes (N = 8192: ~3 min on 16 threads).   usage: truth_check.py <N> [add_noise]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multifidelity_datafusion_gps_amd._lib import Engine
from oracle import gp_oracle as orc, quad_truth
from tests import cases, tolerances as tol

N = int(sys.argv[1])
add_noise = len(sys.argv) > 2 and sys.argv[2] == "add_noise"
rng = np.random.default_rng(N)
X = rng.uniform(size=(N, 4)); Xs = rng.uniform(size=(256, 4))
Y = cases.hf_4d(X); Y = Y - Y.mean()
Xa, Xsa = np.hstack([X, cases.lf_4d(X)[:, None]]), np.hstack([Xs, cases.lf_4d(Xs)[:, None]])
parts, theta = cases.composite(4, 1), np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8])
noise = 1e-6 if add_noise else 0.01 * Y.var()
t = time.time()
tr = quad_truth.evaluate(parts, theta, noise, Xa, Y, Xsa, want_grad=False, want_K=True)
print("N = %d, NARGP composite, noise %.3g: quad precision %.0f s" % (N, noise, time.time() - t), flush=True)
cond = tol.cond_bound(tr["K"], noise); cf = tol.cond_factor(cond); ys = max(1.0, np.abs(Y).max()); kss = tr["K"][0, 0]
e = Engine(0); e.set_data(Xa, Y); e.set_kernel(parts)
nlml = e.eval(theta, noise, 1e-8, want_grad=False)
mean, var = e.predict(Xsa, want_var=True, include_noise=False)
Kh = e.get_K()
st = orc.inference(parts, theta, noise, Xa, Y)
mu, v_exp = orc.predict(parts, theta, noise, Xa, st, Xsa, include_noise=False)
_, v_tri = orc.predict_stable(parts, theta, noise, Xa, st, Xsa, include_noise=False)
Ko = orc.cov(parts, theta, Xa)
tv = np.maximum(tr["var"], 1e-15)
print("cond(Ky) bound %.2e -> tolerance factor %.1f" % (cond, cf))
print("%-34s %12s %12s   (stated tolerance)" % ("error against the quad values", "HIP", "fp64 oracle"))
print("%-34s %12.2e %12.2e   (1e-13 k**)" % ("K, max abs", np.abs(Kh - tr["K"]).max(), np.abs(Ko - tr["K"]).max()))
print("%-34s %12.2e %12.2e   (%.1e)" % ("NLML, relative", abs(nlml - tr["nlml"]) / abs(tr["nlml"]), abs(st["nlml"] - tr["nlml"]) / abs(tr["nlml"]), tol.nlml_rel(cond)))
print("%-34s %12.2e %12.2e   (%.1e)" % ("mean, max abs", np.abs(mean - tr["mean"]).max(), np.abs(np.ravel(mu) - tr["mean"]).max(), tol.PRED_ABS * cf * ys))
print("%-34s %12.2e %12.2e   (%.1e)" % ("variance (triangular), max abs", np.abs(np.maximum(var, 1e-15) - tv).max(), np.abs(np.ravel(v_tri) - tv).max(), tol.PRED_ABS * cf * ys))
print("%-34s %12s %12.2e   (explicit-inverse bound %.1e)" % ("variance (GPy's explicit inverse)", "-", np.abs(np.ravel(v_exp) - tv).max(), tol.explicit_inverse_bound(cond, kss, ys)))
# the few-row forms of the same predict (1-4 rows: few-row panel kernel + VALU product; 5-64: the matrix-pipe product with its partial
# planes; the means in the product launches' mean blocks), against the same quad values
for ns in (1, 2, 4, 8, 16, 32, 64):
    m, v = e.predict(Xsa[:ns], want_var=True, include_noise=False)
    print("%-34s %12.2e %12.2e   (mean / variance, max abs against the quad values; 256-row call: the rows above)"
          % ("predict of the first %d rows" % ns, np.abs(m - tr["mean"][:ns]).max(), np.abs(np.maximum(v, 1e-15) - tv[:ns]).max()))
