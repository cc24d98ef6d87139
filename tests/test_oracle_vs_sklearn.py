"""Independent cross-check of the oracle against scikit-learn's GaussianProcessRegressor.

GPy itself is not available offline (oracle header: PARITY UNPINNED with respect to GPy), but scikit-learn ships a
separately written exact-GP implementation (Rasmussen & Williams alg. 2.1, the same algorithm GPy's
ExactGaussianInference follows).  With matching conventions -- `alpha` = noise + GPy's constant 1e-8 jitter, no target
normalisation, anisotropic length scales with a huge value standing in for the inactive columns of an active_dims
kernel -- covariance, log marginal likelihood, its gradient and the predictive moments must agree.
sklearn's theta is log(parameter); d/dlog(p) = p d/dp is applied to the oracle's gradients."""
import numpy as np
import pytest

from oracle import gp_oracle as orc
from tests import cases

sk = pytest.importorskip("sklearn.gaussian_process")
from sklearn.gaussian_process.kernels import RBF as SkRBF, ConstantKernel as C, Matern  # noqa: E402

HUGE = 1e9   # length scale of an inactive column


def _sk_factor(ktype, var, ls, c0, c1, D):
    scales = np.full(D, HUGE)
    scales[c0:c1] = ls            # one value (isotropic) or one per active column (ARD)
    ktype &= ~orc.ARD
    base = SkRBF(length_scale=scales) if ktype == orc.RBF else Matern(length_scale=scales, nu=1.5 if ktype == orc.MATERN32 else 2.5)
    return C(var) * base


def _sk_kernel(parts, theta, D):
    terms = {}
    theta = np.asarray(theta, dtype=float)
    lay, _ = orc.layout(parts)
    for i, (t, c0, c1, term) in enumerate(parts):
        f = _sk_factor(t, theta[lay[i][0]], theta[lay[i][1]], c0, c1, D)
        terms[term] = f if term not in terms else terms[term] * f
    k = None
    for t in sorted(terms):
        k = terms[t] if k is None else k + terms[t]
    return k


CASES = [n for n in cases.GOLDEN_CASES if n != "rbf_addnoise_n60"]   # (1e-6-noise case: conditioning, not conventions)


@pytest.mark.parametrize("name", CASES)
def test_covariance_nlml_and_prediction_agree_with_sklearn(name):
    c = cases.make_case(name)
    parts, theta, noise, X, Y, Xs = c["parts"], c["theta"], c["noise"], c["X"], c["Y"], c["Xs"]
    D = X.shape[1]
    k = _sk_kernel(parts, theta, D)
    vsum = sum(abs(theta[iv]) for iv, _ in orc.layout(parts)[0])
    np.testing.assert_allclose(orc.cov(parts, theta, X), k(X), rtol=0, atol=1e-10 * vsum)
    np.testing.assert_allclose(orc.cov(parts, theta, X, Xs), k(X, Xs), rtol=0, atol=1e-10 * vsum)
    gpr = sk.GaussianProcessRegressor(kernel=k, alpha=noise + 1e-8, optimizer=None, normalize_y=False).fit(X, Y)
    st = orc.inference(parts, theta, noise, X, Y)
    lml = gpr.log_marginal_likelihood(gpr.kernel_.theta)
    assert -st["nlml"] == pytest.approx(lml, rel=1e-9, abs=1e-8)
    mu_s, sd_s = gpr.predict(Xs, return_std=True)
    mu, var = orc.predict(parts, theta, noise, X, st, Xs, include_noise=False)
    np.testing.assert_allclose(mu, mu_s, rtol=0, atol=1e-7 * max(1.0, np.abs(mu_s).max()))
    np.testing.assert_allclose(var, sd_s ** 2, rtol=0, atol=1e-6 * max(1.0, var.max()))


@pytest.mark.parametrize("name", CASES)
def test_gradient_agrees_with_sklearn(name):
    c = cases.make_case(name)
    parts, theta, noise, X, Y = c["parts"], c["theta"], c["noise"], c["X"], c["Y"]
    D = X.shape[1]
    k = _sk_kernel(parts, theta, D)
    gpr = sk.GaussianProcessRegressor(kernel=k, alpha=noise + 1e-8, optimizer=None).fit(X, Y)
    _, g_sk = gpr.log_marginal_likelihood(gpr.kernel_.theta, eval_gradient=True)
    st = orc.inference(parts, theta, noise, X, Y)
    g = orc.cov_param_grads(parts, theta, X, st["dL_dK"])     # dL/d(var_f), dL/d(len_f) of the log likelihood
    # map sklearn's hyper-parameter vector: per factor [log const, log length_scale[0..D-1]] in kernel-tree order
    names = [h.name for h in gpr.kernel_.hyperparameters]
    sizes = [h.n_elements for h in gpr.kernel_.hyperparameters]
    assert len(names) == 2 * len(parts)
    order = _tree_order(parts)
    lay, _ = orc.layout(parts)
    theta = np.asarray(theta, dtype=float)
    pos = 0
    for slot, i in enumerate(order):
        t, c0, c1, term = parts[i]
        n_c, n_l = sizes[2 * slot], sizes[2 * slot + 1]
        g_var = g_sk[pos]
        g_cols = g_sk[pos + n_c + c0: pos + n_c + c1]            # d/d log(length scale) of every active column
        pos += n_c + n_l
        iv, il = lay[i]
        assert g_var == pytest.approx(theta[iv] * g[iv], rel=1e-7, abs=1e-7)
        if t & orc.ARD:                                          # one lengthscale per column
            np.testing.assert_allclose(g_cols, theta[il] * g[il], rtol=1e-7, atol=1e-7)
        else:                                                    # the isotropic length scale is shared by the active columns
            assert g_cols.sum() == pytest.approx(theta[il][0] * g[il][0], rel=1e-7, abs=1e-7)


def _tree_order(parts):
    """order in which _sk_kernel's tree lists the factors (sklearn enumerates k1's parameters before k2's)."""
    by_term = {}
    for i, p in enumerate(parts):
        by_term.setdefault(p[3], []).append(i)
    return [i for t in sorted(by_term) for i in by_term[t]]
