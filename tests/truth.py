"""Cases and comparison helper of the quad-precision checks (oracle/quad_truth.c): every fp64 path -- the numpy oracle on the CPU
(tests/test_oracle_truth.py), the HIP engine on the GPU (tests/test_gpu_truth.py) -- against the SAME 113-bit evaluation of the same
quantities on the same fp64 inputs.  Where the other parity tests say "two fp64 evaluations agree to the stated tolerance", these say
"each is within the stated tolerance of the true value" (times the cond(Ky) factor of tests/tolerances.py where cond > 1e7).
"""
import numpy as np

from tests import cases
from tests import tolerances as tol


def _aug(f_lf):
    return lambda X: np.hstack([X, f_lf(X)[:, None]])


# name -> N, columns, kernel, theta, target, augmentation, noise (relative to var(y) unless `abs_noise`), what to compute
TRUTH_CASES = {
    "rbf_3d_n300": dict(N=300, d=3, parts=cases.single(cases.RBF, 3), theta=[1.1, 0.3], f=cases.hf_3d),
    "nargp_4d_n400": dict(N=400, d=4, parts=cases.composite(4, 1), theta=[1.2, 1.1, 0.9, 0.6, 0.4, 0.8], f=cases.hf_4d, aug=_aug(cases.lf_4d)),
    "matern52_ard_3d_n350": dict(N=350, d=3, parts=cases.single(cases.M52 | cases.ARD, 3), theta=[0.9, 0.5, 0.8, 0.35], f=cases.hf_3d),
    "mixed_composite_4d_n320": dict(N=320, d=4, parts=cases.composite(4, 1, cases.M52, cases.RBF, cases.M32),
                                    theta=[1.1, 1.3, 0.7, 0.8, 0.5, 0.9], f=cases.hf_4d, aug=_aug(cases.lf_4d)),
    "nargp_ard_4d_n260": dict(N=260, d=4, parts=cases.composite(4, 1, cases.RBF, cases.RBF | cases.ARD, cases.RBF | cases.ARD),
                              theta=[1.2, 1.1, 0.9, 0.6, 0.7, 0.5, 0.8, 0.4, 0.8, 0.9, 0.6, 0.7], f=cases.hf_4d, aug=_aug(cases.lf_4d)),
    # src/MFDataFusion.py:154-155: the adaptation loop's add_noise regime, sigma_n^2 = 1e-6 (cond(Ky) ~ 1e9 .. 1e10)
    "add_noise_rbf_2d_n256": dict(N=256, d=2, parts=cases.single(cases.RBF, 2), theta=[1.0, 0.4], f=cases.hf_2d, abs_noise=1e-6),
    "add_noise_nargp_2d_n200": dict(N=200, d=2, parts=cases.composite(2, 1), theta=[1.0, 1.5, 0.9, 0.5, 0.3, 0.6], f=cases.hf_2d,
                                    aug=_aug(cases.lf_2d), abs_noise=1e-6),
    # smooth kernel, almost no noise: cond(Ky) ~ 3e9 .. 2e10 -- where the cond factor of tests/tolerances.py is earned (or not)
    "ill_conditioned_rbf_2d_n300": dict(N=300, d=2, parts=cases.single(cases.RBF, 2), theta=[1.0, 0.8], f=cases.hf_2d, abs_noise=1e-7),
    "ill_conditioned_matern52_3d_n400": dict(N=400, d=3, parts=cases.single(cases.M52, 3), theta=[2.0, 1.5], f=cases.hf_3d, abs_noise=1e-8),
}
# sizes only the GPU suite runs (quad-precision time grows with N^3: 7 s at N = 1000, about a minute at 2100 on 8 threads)
GPU_TRUTH_CASES = {
    "rbf_3d_n1000": dict(N=1000, d=3, parts=cases.single(cases.RBF, 3), theta=[1.1, 0.3], f=cases.hf_3d),
    "nargp_4d_n2100": dict(N=2100, d=4, parts=cases.composite(4, 1), theta=[1.2, 1.1, 0.9, 0.6, 0.4, 0.8], f=cases.hf_4d, aug=_aug(cases.lf_4d)),
    "matern32_4d_n1500": dict(N=1500, d=4, parts=cases.single(cases.M32, 4), theta=[1.3, 0.7], f=cases.hf_4d),
    "rbf_3d_n4096_no_gradient": dict(N=4096, d=3, parts=cases.single(cases.RBF, 3), theta=[1.1, 0.25], f=cases.hf_3d, grad=False),
}
ALL = dict(TRUTH_CASES, **GPU_TRUTH_CASES)


def make(name, n_star=64):
    c = ALL[name]
    rng = np.random.default_rng(sum(map(ord, name)))
    X = rng.uniform(size=(c["N"], c["d"]))
    Xs = rng.uniform(size=(n_star, c["d"]))
    Y = c["f"](X)
    Y = Y - Y.mean()
    if "aug" in c:
        X, Xs = c["aug"](X), c["aug"](Xs)
    noise = c["abs_noise"] if "abs_noise" in c else 0.01 * Y.var()
    return dict(parts=c["parts"], theta=np.array(c["theta"], dtype=np.float64), noise=float(noise), X=X, Y=Y, Xs=Xs,
                want_grad=c.get("grad", True))


def truth_of(case):
    from oracle import quad_truth
    return quad_truth.evaluate(case["parts"], case["theta"], case["noise"], case["X"], case["Y"], case["Xs"],
                               want_grad=case["want_grad"], want_K=True)


def check_against_truth(label, case, tr, nlml=None, grad=None, mean=None, var=None, K=None, var_explicit=None):
    """asserts each given fp64 result against the quad-precision one at the stated tolerances (x the cond factor); records
    error / tolerance.  `var` is a latent variance computed through the triangular factor, `var_explicit` one computed the way GPy does
    (explicit inverse: its own error bound, tolerances.explicit_inverse_bound)."""
    cond = tol.cond_bound(tr["K"], case["noise"])
    cf = tol.cond_factor(cond)
    ys = max(1.0, float(np.abs(case["Y"]).max()))
    kss = float(tr["K"][0, 0])
    if K is not None:
        err = np.abs(K - tr["K"]).max()
        tol._record(label, "K_abs_over_1e-13", err / (1e-13 * kss))
        assert err <= 1e-13 * kss, err
    if nlml is not None:
        # an NLML near zero is a sum of terms that are not: the relative bound is on the terms' magnitude (tools/fuzz_parity.py)
        scale = max(abs(tr["nlml"]), 0.5 * (case["X"].shape[0] * np.log(2 * np.pi) + abs(tr["logdet"])))
        err = abs(nlml - tr["nlml"]) / scale
        tol._record(label, "nlml_rel", err / tol.nlml_rel(cond))
        assert err <= tol.nlml_rel(cond), (nlml, tr["nlml"], err)
    if grad is not None:
        tol.check_grad(grad, tr["grad"], rel=tol.GRAD_REL * cf, label=label)
    if mean is not None:
        tol.check_pred(mean, tr["mean"], ys, tol.PRED_ABS * cf, label=label, what="mean")
    if var is not None:
        tol.check_pred(np.maximum(var, 1e-15), np.maximum(tr["var"], 1e-15), ys, tol.PRED_ABS * cf, label=label, what="var")
    if var_explicit is not None:
        bound = tol.explicit_inverse_bound(cond, kss, ys)
        err = np.abs(np.maximum(var_explicit, 1e-15) - np.maximum(tr["var"], 1e-15)).max()
        tol._record(label, "var_explicit_inverse_over_bound", err / bound)
        assert err <= bound, (err, bound)
    tol._record(label, "cond_factor", cf)
    return cf


QUAD_MAX_ROWS = 4096      # quad-precision Cholesky: 2 s at 1024 rows, 25 s at 4096, 194 s at 8192 (16 threads)


def check_add_noise_state(label, parts, theta, noise, X, Y, Xs, nlml, mean, var, jitter=1e-8, quad_rows=256, quad_max_rows=None):
    """One fitted / appended state in the add_noise regime (sigma_n^2 = 1e-6, src/MFDataFusion.py:154-155) -- `nlml`, `mean`, `var`
    (noise INCLUDED, as MultifidelityDataFusion.predict returns it) from the HIP path at (parts, theta, noise) on the rows (X, Y):

      (1) against the QUAD-PRECISION values at the STATED add_noise tolerances (NLML rel 1e-7, mean / variance
          1e-7 max(1, |y|_inf)) wherever N <= QUAD_MAX_ROWS (on the first `quad_rows` test rows: N^2 quad flops per row);
      (2) against the fp64 oracle -- BOTH predictive forms: the triangular one and GPy's explicit inverse, which is what
          src/MFDataFusion.py:156 returns -- at tolerances DERIVED from the case's cond(Ky) bound (tests/tolerances.py:
          fp64_pair_nlml_rel, fp64_pair_pred_abs, explicit_inverse_bound): two rounded evaluations, each carrying O(eps cond).
    Returns the oracle state."""
    from oracle import gp_oracle as orc
    Y = np.asarray(Y, dtype=np.float64).reshape(-1)
    mean, var = np.asarray(mean).reshape(-1), np.asarray(var).reshape(-1)
    ys = max(1.0, float(np.abs(Y).max()))
    st = orc.inference(parts, theta, noise, X, Y, want_grad=False, const_jitter=jitter)
    cond = tol.cond_bound(st["K"], noise, jitter)
    kss = float(orc.cov_diag(parts, theta, 1)[0])
    mu, v_inv = orc.predict(parts, theta, noise, X, st, Xs)
    _, v_tri = orc.predict_stable(parts, theta, noise, X, st, Xs)
    tol.check_nlml(nlml, st["nlml"], rel=tol.fp64_pair_nlml_rel(cond), label=label)
    tol.check_pred(mean, mu, 1.0, tol.fp64_pair_pred_abs(cond, ys), label=label, what="mean_vs_fp64_oracle")
    tol.check_pred(var, v_tri, 1.0, tol.fp64_pair_pred_abs(cond, ys), label=label, what="var_vs_fp64_triangular")
    tol.check_pred(var, v_inv, 1.0, tol.explicit_inverse_bound(cond, kss, ys, base=tol.PRED_ABS_ADDNOISE), label=label,
                   what="var_vs_fp64_explicit_inverse")
    tol._record(label, "cond_bound", cond)
    tol._record(label, "nlml_pair_err_over_eps_cond", abs(nlml - st["nlml"]) / abs(st["nlml"]) / (np.finfo(float).eps * cond))
    tol._record(label, "mean_pair_err_over_eps_cond", np.abs(mean - mu).max() / ys / (np.finfo(float).eps * cond))
    if X.shape[0] <= (QUAD_MAX_ROWS if quad_max_rows is None else min(QUAD_MAX_ROWS, quad_max_rows)):
        from oracle import quad_truth
        q = slice(0, min(quad_rows, len(mean)))
        tr = quad_truth.evaluate(parts, theta, noise, X, Y, np.ascontiguousarray(Xs[q]), jitter=jitter, want_grad=False)
        scale = max(abs(tr["nlml"]), 0.5 * (X.shape[0] * np.log(2 * np.pi) + abs(tr["logdet"])))
        err = abs(nlml - tr["nlml"]) / scale
        tol._record(label, "nlml_vs_quad_over_1e-7", err / tol.NLML_REL_ADDNOISE)
        assert err <= tol.NLML_REL_ADDNOISE, (nlml, tr["nlml"], err)
        tol.check_pred(mean[q], tr["mean"], ys, tol.PRED_ABS_ADDNOISE, label=label, what="mean_vs_quad")
        tol.check_pred(np.maximum(var[q] - noise, 1e-15), np.maximum(tr["var"], 1e-15), ys, tol.PRED_ABS_ADDNOISE, label=label,
                       what="var_vs_quad")
    return st
