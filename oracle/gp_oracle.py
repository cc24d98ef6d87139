"""CPU oracle: numpy/LAPACK restatement of the exact-GP arithmetic the reference delegates to GPy.

TEST INFRASTRUCTURE ONLY.  Nothing in the product (multifidelity_datafusion_gps_amd/) imports this
module; only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg do, and only as the
checker / the timed CPU comparator.

PARITY UNPINNED.  The arithmetic lives in third-party `GPy==1.9.9` + `paramz==0.9.5`
(/root/reference/requirements.txt:11,25), which is neither vendored under /root/reference nor
installable offline, and the reference's own tests hold no golden vector for K, L, alpha, NLML,
mean or variance.  This file therefore restates GPy-1.9.9's published
algorithm **from knowledge of that version** (statements tagged [GPy-recall]) and is pinned only by
self-consistency checks (finite-difference gradients, closed forms for N = 1, 2, invariances) in
tests/test_oracle.py plus the golden vectors it generated itself (tests/golden/), and cross-checked against an
independently written exact-GP implementation available offline (scikit-learn's GaussianProcessRegressor:
covariance, log marginal likelihood, gradient, predictive moments -- tests/test_oracle_vs_sklearn.py).  That
check pins the MATH, not GPy's conventions (jitter constant, transforms, optimiser controls), which stay
[GPy-recall].

Reference call sites this follows (what the reference ASKS of the engine):
  * kernels: src/abstractMFGP.py:59-60 (GPy.kern.RBF(D)), :62-80 (kern1*kern2 + kern3, active_dims)
  * model / inference: src/MFDataFusion.py:93-98, src/abstractMFGP.py:100-102 (GPRegression)
  * hyper-parameter recipe: src/abstractMFGP.py:131-137 (ARD)
  * prediction: src/MFDataFusion.py:154-156, src/abstractMFGP.py:104

Kernel description (same POD layout as include/mfgp.h): parts = [(type, col_begin, col_end, term)],
theta = [var_0, len_0, var_1, len_1, ...];  K = sum_terms prod_{f in term} k_f.  A factor whose type carries the ARD flag has one
lengthscale per active column, in column order, in place of len_f (theta = [var_0, len_0_0 .. len_0_{k-1}, var_1, ...]).
"""
import numpy as np
from scipy.linalg import lapack

RBF, MATERN32, MATERN52 = 0, 1, 2
ARD = 0x100            # flag on the type: GPy's ARD=True (Stationary with one lengthscale per input dimension)
LOG_2_PI = np.log(2.0 * np.pi)


def layout(parts):
    """-> [(index of variance_f, slice of its lengthscale(s))] and the number of kernel parameters P"""
    out, i = [], 0
    for ktype, c0, c1, _ in parts:
        nl = (c1 - c0) if (ktype & ARD) else 1
        out.append((i, slice(i + 1, i + 1 + nl)))
        i += 1 + nl
    return out, i


# ------------------------------------------------------------------------------------------------
# stationary kernels  [GPy-recall: GPy/kern/src/stationary.py, rbf.py]
# ------------------------------------------------------------------------------------------------
def unscaled_dist(X, X2=None):
    """Stationary._unscaled_dist: r = sqrt(clip(|x|^2 + |x'|^2 - 2 x.x', 0)), diagonal forced to 0."""
    if X2 is None:
        Xsq = np.sum(np.square(X), 1)
        r2 = -2.0 * X.dot(X.T) + (Xsq[:, None] + Xsq[None, :])
        r2[np.diag_indices(X.shape[0])] = 0.0
        r2 = np.clip(r2, 0, np.inf)
        return np.sqrt(r2)
    X1sq = np.sum(np.square(X), 1)
    X2sq = np.sum(np.square(X2), 1)
    r2 = -2.0 * X.dot(X2.T) + (X1sq[:, None] + X2sq[None, :])
    r2 = np.clip(r2, 0, np.inf)
    return np.sqrt(r2)


def k_of_r(ktype, variance, r):
    if ktype == RBF:       # RBF.K_of_r
        return variance * np.exp(-0.5 * r ** 2)
    if ktype == MATERN32:  # Matern32.K_of_r
        return variance * (1.0 + np.sqrt(3.0) * r) * np.exp(-np.sqrt(3.0) * r)
    if ktype == MATERN52:  # Matern52.K_of_r
        return variance * (1 + np.sqrt(5.0) * r + 5.0 / 3 * r ** 2) * np.exp(-np.sqrt(5.0) * r)
    raise ValueError("unknown kernel type")


def dk_dr(ktype, variance, r):
    if ktype == RBF:       # RBF.dK_dr = -r K
        return -r * k_of_r(RBF, variance, r)
    if ktype == MATERN32:
        return -3.0 * variance * r * np.exp(-np.sqrt(3.0) * r)
    if ktype == MATERN52:
        return variance * (10.0 / 3 * r - 5.0 * r - 5.0 * np.sqrt(5.0) / 3 * r ** 2) * np.exp(-np.sqrt(5.0) * r)
    raise ValueError("unknown kernel type")


def _factor_r(part, lengthscale, X, X2):
    ktype, c0, c1, _ = part
    Xa = X[:, c0:c1]
    X2a = None if X2 is None else X2[:, c0:c1]
    lengthscale = np.asarray(lengthscale, dtype=np.float64).reshape(-1)
    if ktype & ARD:                                  # Stationary._scaled_dist, ARD branch: distances of the rescaled inputs
        return unscaled_dist(Xa / lengthscale, None if X2a is None else X2a / lengthscale)
    return unscaled_dist(Xa, X2a) / lengthscale[0]   # Stationary._scaled_dist (non-ARD)


def _terms(parts):
    out = {}
    for f, p in enumerate(parts):
        out.setdefault(p[3], []).append(f)
    return [out[t] for t in sorted(out)]


def cov(parts, theta, X, X2=None):
    """K(X, X2) for the sum-of-products structure (Prod.K = product of part Ks, Add.K = sum)."""
    K = 0.0
    lay, _ = layout(parts)
    theta = np.asarray(theta, dtype=np.float64)
    for fs in _terms(parts):
        prod = 1.0
        for f in fs:
            iv, il = lay[f]
            prod = prod * k_of_r(parts[f][0] & ~ARD, theta[iv], _factor_r(parts[f], theta[il], X, X2))
        K = K + prod
    return K


def cov_diag(parts, theta, n):
    """Kdiag: stationary kernels return their variance; Prod multiplies, Add sums."""
    kss = 0.0
    lay, _ = layout(parts)
    for fs in _terms(parts):
        prod = 1.0
        for f in fs:
            prod *= theta[lay[f][0]]
        kss += prod
    return np.full(n, kss)


def cov_param_grads(parts, theta, X, dL_dK):
    """update_gradients_full for every factor: returns d(sum dL_dK o K)/d theta (same layout as theta).

    RBF/Stationary: variance.gradient = sum(K o dL_dK)/variance ;
                    lengthscale.gradient = -sum(dL_dr o r)/lengthscale , dL_dr = dK_dr o dL_dK.
    Prod: each part sees dL_dK multiplied by the K of the other parts of its product.
    ARD (Stationary._lengthscale_grads_pure): lengthscale_q.gradient = -sum_ij (dL_dr / r)_ij (x_iq - x_jq)^2 / l_q^3,
    with 1/r := 0 where r = 0 (Stationary._inv_dist).
    """
    theta = np.asarray(theta, dtype=np.float64)
    g = np.zeros(len(theta))
    lay, _ = layout(parts)
    for fs in _terms(parts):
        Ks, rs = {}, {}
        for f in fs:
            iv, il = lay[f]
            rs[f] = _factor_r(parts[f], theta[il], X, None)
            Ks[f] = k_of_r(parts[f][0] & ~ARD, theta[iv], rs[f])
        for f in fs:
            iv, il = lay[f]
            other = 1.0
            for h in fs:
                if h != f:
                    other = other * Ks[h]
            dl = dL_dK * other
            g[iv] = np.sum(Ks[f] * dl) / theta[iv]
            dL_dr = dk_dr(parts[f][0] & ~ARD, theta[iv], rs[f]) * dl
            if parts[f][0] & ARD:
                with np.errstate(divide="ignore", invalid="ignore"):
                    tmp = np.where(rs[f] > 0, dL_dr / rs[f], 0.0)
                c0 = parts[f][1]
                for q, ell in enumerate(theta[il]):
                    dq = X[:, c0 + q:c0 + q + 1] - X[:, c0 + q:c0 + q + 1].T
                    g[il.start + q] = -np.sum(tmp * np.square(dq)) / ell ** 3
            else:
                g[il.start] = -np.sum(dL_dr * rs[f]) / theta[il.start]
    return g


# ------------------------------------------------------------------------------------------------
# linear algebra  [GPy-recall: GPy/util/linalg.py jitchol, pdinv, dpotrs, dpotri, dtrtri]
# ------------------------------------------------------------------------------------------------
def jitchol(A, maxtries=5):
    """Lower Cholesky with GPy's jitter-retry policy: jitter = mean(diag)*1e-6, x10 per retry, 5 tries."""
    A = np.ascontiguousarray(A)
    L, info = lapack.dpotrf(A, lower=1)
    if info == 0:
        return L, 0.0
    diagA = np.diag(A)
    if np.any(diagA <= 0.0):
        raise np.linalg.LinAlgError("not pd: non-positive diagonal elements")
    jitter = diagA.mean() * 1e-6
    num_tries = 1
    while num_tries <= maxtries and np.isfinite(jitter):
        L, info = lapack.dpotrf(np.ascontiguousarray(A + np.eye(A.shape[0]) * jitter), lower=1)
        if info == 0:
            return L, jitter
        jitter *= 10
        num_tries += 1
    raise np.linalg.LinAlgError("not positive definite, even with jitter.")


def pdinv(A):
    """Returns (Ai, L, Li, logdet) exactly like GPy.util.linalg.pdinv."""
    L, _ = jitchol(A)
    logdet = 2.0 * np.sum(np.log(np.diag(L)))
    Li, _ = lapack.dtrtri(L, lower=1)
    Ai, _ = lapack.dpotri(L, lower=1)
    Ai = np.tril(Ai) + np.tril(Ai, -1).T   # symmetrify
    return Ai, L, Li, logdet


# ------------------------------------------------------------------------------------------------
# exact Gaussian inference  [GPy-recall: inference/latent_function_inference/exact_gaussian_inference.py]
# ------------------------------------------------------------------------------------------------
def inference(parts, theta, noise, X, Y, want_grad=True, const_jitter=1e-8):
    """One objective(+gradient) evaluation.  Y is (N,) or (N,1).  Returns a dict."""
    Y = np.asarray(Y, dtype=np.float64).reshape(-1, 1)
    N = X.shape[0]
    K = cov(parts, theta, X)
    Ky = K.copy()
    Ky[np.diag_indices(N)] += noise + const_jitter
    Wi, LW, LWi, W_logdet = pdinv(Ky)
    alpha, _ = lapack.dpotrs(LW, Y, lower=1)
    log_marginal = 0.5 * (-Y.size * LOG_2_PI - Y.shape[1] * W_logdet - np.sum(alpha * Y))
    out = dict(K=K, L=LW, Linv=LWi, Kinv=Wi, alpha=alpha[:, 0], logdet=W_logdet, nlml=-log_marginal)
    if want_grad:
        dL_dK = 0.5 * (alpha.dot(alpha.T) - Y.shape[1] * Wi)
        g_kern = cov_param_grads(parts, theta, X, dL_dK)
        g_noise = np.sum(np.diag(dL_dK))              # Gaussian.exact_inference_gradients
        out["dL_dK"] = dL_dK
        out["grad"] = -np.concatenate([g_kern, [g_noise]])   # objective = -log marginal
    return out


def predict(parts, theta, noise, X, state, Xnew, include_noise=True):
    """Posterior._raw_predict + Gaussian.predictive_values (mean, variance incl. noise).

    [GPy-recall] woodbury_inv = dpotri(woodbury_chol) -> var = Kdiag - sum((Kinv Kx) o Kx, 0),
    clipped below at 1e-15, + likelihood variance.
    """
    Kx = cov(parts, theta, X, Xnew)                # (N, N*)
    mu = Kx.T.dot(state["alpha"])
    Kxx = cov_diag(parts, theta, Xnew.shape[0])
    var = Kxx - np.sum(np.dot(state["Kinv"].T, Kx) * Kx, 0)
    var = np.clip(var, 1e-15, np.inf)
    if include_noise:
        var = var + noise
    return mu, var


def predict_stable(parts, theta, noise, X, state, Xnew, include_noise=True):
    """Same quantity through the triangular factor: var = Kdiag - |L^-1 kx|^2 (what the HIP path computes).

    Algebraically identical to `predict`; better conditioned.  Used by the parity tests as the tighter
    comparator in the add_noise regime (sigma_n^2 = 1e-6), where the explicit-inverse form of GPy loses
    digits to cancellation itself.
    """
    Kx = cov(parts, theta, X, Xnew)
    mu = Kx.T.dot(state["alpha"])
    V, _ = lapack.dtrtrs(state["L"], Kx, lower=1)
    var = cov_diag(parts, theta, Xnew.shape[0]) - np.sum(V * V, 0)
    var = np.clip(var, 1e-15, np.inf)
    if include_noise:
        var = var + noise
    return mu, var


# ------------------------------------------------------------------------------------------------
# positivity transform  [GPy-recall: paramz.transformations.Logexp]
# ------------------------------------------------------------------------------------------------
_LIM_VAL = 36.0
_LOG_LIM_VAL = np.log(np.finfo(np.float64).max)   # paramz: _log_lim_val = log(DBL_MAX) ~ 709.78


def logexp_f(x):
    """optimizer space -> positive parameter: log(1 + e^x); linear above _lim_val = 36, the argument of exp clipped
    to [-log(DBL_MAX), 36].  paramz 0.9.5 transformations.Logexp.f ends in `#+ epsilon`: the epsilon term is
    commented out upstream, so none is added here [GPy-recall]."""
    x = np.asarray(x, dtype=np.float64)
    return np.where(x > _LIM_VAL, x, np.log1p(np.exp(np.clip(x, -_LOG_LIM_VAL, _LIM_VAL))))


def logexp_finv(f):
    f = np.asarray(f, dtype=np.float64)
    return np.where(f > _LIM_VAL, f, np.log(np.expm1(f)))


def logexp_gradfactor(f, df):
    """d objective / d x = df * (1 - e^-f)  (1 above the limit)."""
    f = np.asarray(f, dtype=np.float64)
    return df * np.where(f > _LIM_VAL, 1.0, -np.expm1(-f))


def objective_transformed(parts, x, X, Y, fixed_noise=None):
    """NLML and gradient in the optimizer (softplus) space; x = [theta..., noise] or [theta...] if noise fixed."""
    p = logexp_f(x)
    if fixed_noise is None:
        theta, noise = p[:-1], p[-1]
    else:
        theta, noise = p, fixed_noise
    st = inference(parts, theta, noise, X, Y, want_grad=True)
    g = st["grad"]
    if fixed_noise is not None:
        g = g[:-1]
    return st["nlml"], logexp_gradfactor(p, g)
