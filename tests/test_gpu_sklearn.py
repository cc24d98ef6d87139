"""A second, independent checker ON THE GPU OUTPUTS: the HIP engine (through the C-ABI) against scikit-learn's
GaussianProcessRegressor -- a separately written exact GP (Rasmussen & Williams alg. 2.1) that shares no code with
oracle/gp_oracle.py -- at N ~ 2000 for an RBF, a Matern and the NARGP composite k1*k2 + k3 (src/abstractMFGP.py:62-80).

The oracle is "parity unpinned" with respect to GPy (its header); tests/test_oracle_vs_sklearn.py checks
the ORACLE against scikit-learn on the CPU.  Here scikit-learn checks the product itself: covariance matrix, log marginal
likelihood, its gradient for every hyper-parameter, predictive mean and latent variance.  Conventions as there: `alpha` =
noise + GPy's constant 1e-8 jitter, no target normalisation, a huge length scale for the columns an active_dims factor ignores;
scikit-learn differentiates with respect to log(parameter): d/dlog(p) = p d/dp.
"""
import numpy as np
import pytest

from tests import cases
from tests import tolerances as tol
from tests.test_oracle_vs_sklearn import _sk_kernel, _tree_order

sk = pytest.importorskip("sklearn.gaussian_process")

pytestmark = pytest.mark.gpu


def _layout(parts):
    out, i = [], 0
    for t, c0, c1, _ in parts:
        nl = (c1 - c0) if (t & cases.ARD) else 1
        out.append((i, slice(i + 1, i + 1 + nl)))
        i += 1 + nl
    return out


CASES = {
    "rbf_3d_n2000": dict(N=2000, d=3, parts=cases.single(cases.RBF, 3), theta=[1.1, 0.3], f=cases.hf_3d),
    "matern52_ard_3d_n1900": dict(N=1900, d=3, parts=cases.single(cases.M52 | cases.ARD, 3), theta=[0.9, 0.5, 0.8, 0.35], f=cases.hf_3d),
    "matern32_4d_n1800": dict(N=1800, d=4, parts=cases.single(cases.M32, 4), theta=[1.3, 0.7], f=cases.hf_4d),
    "nargp_composite_4d_n2048": dict(N=2048, d=4, parts=cases.composite(4, 1), theta=[1.2, 1.1, 0.9, 0.6, 0.4, 0.8], f=cases.hf_4d,
                                     aug=cases.lf_4d),
    "nargp_composite_matern_4d_n1700": dict(N=1700, d=4, parts=cases.composite(4, 1, cases.M52, cases.RBF, cases.M32),
                                            theta=[1.1, 1.3, 0.7, 0.8, 0.5, 0.9], f=cases.hf_4d, aug=cases.lf_4d),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_outputs_against_scikit_learn(engine, name):
    c = CASES[name]
    rng = np.random.default_rng(sum(map(ord, name)))
    N, d, parts, theta = c["N"], c["d"], c["parts"], np.array(c["theta"], dtype=np.float64)
    X = rng.uniform(size=(N, d))
    Xs = rng.uniform(size=(150, d))
    Y = c["f"](X)
    if "aug" in c:
        X = np.hstack([X, c["aug"](X)[:, None]])
        Xs = np.hstack([Xs, c["aug"](Xs)[:, None]])
    Y = Y - Y.mean()                       # scikit-learn's prior mean is zero, as GPy's (normalize_y=False)
    noise = 0.01 * Y.var()
    D = X.shape[1]

    engine.set_data(X, Y)
    engine.set_kernel(parts)
    nlml, grad = engine.eval(theta, noise, 1e-8, want_grad=True)
    mean, var = engine.predict(Xs, want_var=True, include_noise=False)
    K = engine.get_K()

    k = _sk_kernel(parts, theta, D)
    Ksk = k(X)
    vsum = sum(theta[iv] for iv, _ in _layout(parts))
    assert np.abs(K - Ksk).max() <= 1e-12 * vsum ** 2, np.abs(K - Ksk).max()
    gpr = sk.GaussianProcessRegressor(kernel=k, alpha=noise + 1e-8, optimizer=None, normalize_y=False).fit(X, Y)
    lml, g_sk = gpr.log_marginal_likelihood(gpr.kernel_.theta, eval_gradient=True)
    cf = tol.cond_factor(tol.cond_bound(Ksk, noise))
    tol.check_nlml(nlml, -lml, rel=tol.nlml_rel(tol.cond_bound(Ksk, noise)), label="sklearn/" + name)

    # scikit-learn's hyper-parameter vector: per factor [log const, log length_scale[0..D-1]] in kernel-tree order
    sizes = [h.n_elements for h in gpr.kernel_.hyperparameters]
    assert len(sizes) == 2 * len(parts)
    lay = _layout(parts)
    want = np.zeros(len(theta))            # d(-lml)/d theta in the engine's layout, from scikit-learn's d lml / d log theta
    pos = 0
    for slot, i in enumerate(_tree_order(parts)):
        t, c0, c1, _ = parts[i]
        n_c, n_l = sizes[2 * slot], sizes[2 * slot + 1]
        iv, il = lay[i]
        want[iv] = -g_sk[pos] / theta[iv]
        cols = g_sk[pos + n_c + c0: pos + n_c + c1]
        if t & cases.ARD:
            want[il] = -cols / theta[il]
        else:
            want[il] = -cols.sum() / theta[il][0]      # the isotropic length scale is shared by the active columns
        pos += n_c + n_l
    tol.check_grad(grad[:-1], want, rel=tol.GRAD_REL * cf, label="sklearn/" + name)     # (alpha is not a scikit-learn hyper-parameter)

    mu_s, sd_s = gpr.predict(Xs, return_std=True)
    ys = max(1.0, np.abs(Y).max())
    tol.check_pred(mean, mu_s, ys, tol.PRED_ABS * cf, label="sklearn/" + name, what="mean")
    # scikit-learn returns a standard deviation: var = sd^2 carries 2 sd * d(sd); compare on the variance scale it supports
    tol.check_pred(var, sd_s ** 2, ys, tol.PRED_ABS * cf, label="sklearn/" + name, what="var")
    tol._record("sklearn/" + name, "cond_factor", cf)
