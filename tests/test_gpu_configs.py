"""GPU tests of BASELINE.json configurations 3, 4 and 5 at their stated sizes.

  cfg3: 2-fidelity NARGP, 4-D, N_lf = 16384 (single RBF) / N_hf = 4096 (composite kernel on D = 5)
  cfg4: three chained fidelity levels, 2-D, N = 8192 per level, + the row-block K build in 8 blocks
  cfg5: NARGP + entropy-reduction adaptation with add_noise=True, N_hf growing across 128-row boundaries

Where a host O(N^3) run is affordable (N <= 8192) the checker is the oracle at the FITTED hyper-parameters; the
N = 16384 level is checked through size-independent properties (sampled residual of Ky alpha = y, permutation
invariance, directional central difference).  Flows follow /root/reference/tests/utils.py:38-47,75-86 and
/root/reference/src/abstractMFGP.py:317-359.
"""
import numpy as np
import pytest

from oracle import gp_oracle as orc
from tests import cases
from tests import tolerances as tol
from tests import truth

pytestmark = pytest.mark.gpu


def _check_fitted_level(model, Xs, mean, var, label):
    """The fitted high-fidelity level against the oracle AT the fitted hyper-parameters, with the stated tolerances
    (tests/tolerances.py) times the conditioning factor of the fitted Ky (the optimiser is free to drive the noise variance
    down: 1 up to cond ~ 1e7, linear beyond).  The variance is compared with BOTH predictive forms of the oracle: GPy's
    explicit-inverse one (what the reference's predict returns, src/MFDataFusion.py:156) and the triangular one."""
    parts, theta, noise = _theta_noise(model)
    Xa, Y = model.hf_model.X, model.hf_Y[:, 0]
    st = orc.inference(parts, theta, noise, Xa, Y)
    cf = tol.cond_factor(tol.cond_bound(st["K"], noise))
    Xsa = model._augment_data(Xs)
    mu, v_inv = orc.predict(parts, theta, noise, Xa, st, Xsa)
    _, v_tri = orc.predict_stable(parts, theta, noise, Xa, st, Xsa)
    tol.check_nlml(model.hf_model.objective_function(), st["nlml"], rel=tol.nlml_rel(tol.cond_bound(st["K"], noise)), label=label)
    g = model.hf_model._engine.eval(theta, noise, model.hf_model._jitter_used, want_grad=True)[1]
    tol.check_grad(g, st["grad"], rel=tol.GRAD_REL * cf, label=label)
    ys = np.abs(Y).max()
    tol.check_pred(mean, mu, ys, tol.PRED_ABS * cf, label=label, what="mean")
    tol.check_pred(var, v_tri, ys, tol.PRED_ABS * cf, label=label, what="var_triangular")
    cond = tol.cond_bound(st["K"], noise)
    kss = orc.cov_diag(parts, theta, 1)[0]
    tol.check_pred(var, v_inv, 1.0, tol.explicit_inverse_bound(cond, kss, ys), label=label, what="var_explicit_inverse")
    tol._record(label, "cond_factor", cf)
    tol._record(label, "cond_bound", cond)
    tol._record(label, "explicit_inverse_err_over_eps_cond_kss",
                np.abs(var - v_inv).max() / (np.finfo(float).eps * cond * kss))
    return st


def col(f):
    return lambda x: f(x)[:, None]


def _theta_noise(model):
    parts, plist = model.kernel.engine_parts()
    theta = np.array([q.value for v, ls in plist for q in [v] + ls])
    return parts, theta, model.hf_model.likelihood.variance.value


def _budgeted(evals=4, restarts=2, conc=1):
    """NARGP with a fixed evaluation budget per L-BFGS-B run.  A subclass, not attribute assignment on the instance:
    the data-driven low-fidelity level is fitted inside the constructor and must see the budget too."""
    import multifidelity_datafusion_gps_amd as mf

    class BudgetNARGP(mf.NARGP):
        lf_max_iters = first_run_max_iters = restart_max_iters = evals
        eval_cap = evals
        num_restarts = restarts
        restart_concurrency = conc
    return BudgetNARGP


# ---------------------------------------------------------------------------------------------------------------
# cfg3
# ---------------------------------------------------------------------------------------------------------------
def test_cfg3_lf_level_n16384_properties(engine):
    """The N_lf = 16384 level of cfg3 (single RBF over d = 4; K = 2.15 GB, 128 leaf blocks, 32 macro panels): one
    objective+gradient evaluation and a prediction, checked without any O(N^3) host work."""
    rng = np.random.default_rng(2)
    N = 16384
    X = rng.uniform(size=(N, 4))
    Y = cases.lf_4d(X)
    parts, theta, noise = cases.single(cases.RBF, 4), np.array([1.3, 0.45]), 0.01 * Y.var()
    engine.set_data(X, Y)
    engine.set_kernel(parts)
    nlml, grad = engine.eval(theta, noise)
    t = engine.timings()
    print("cfg3 LF N=16384: total %.1f ms, cholinv %.1f ms, kinv %.1f ms, kbuild %.3f ms"
          % (t["total_ms"], t["cholinv_ms"], t["kinv_ms"], t["kbuild_ms"]))
    assert np.isfinite(nlml) and np.all(np.isfinite(grad))
    alpha = engine.get_alpha()
    # (1) alpha solves Ky alpha = y: residual on sampled rows, the rows of Ky from the oracle's kernel
    rows = rng.choice(N, 96, replace=False)
    res = orc.cov(parts, theta, X[rows], X) @ alpha + (noise + 1e-8) * alpha[rows] - Y[rows]
    assert np.abs(res).max() <= 1e-8 * np.abs(Y).max()
    # (2) posterior mean at fresh points = k(x*, X) alpha with the oracle's kernel rows; variance within [0, sigma^2]
    Xs = rng.uniform(size=(200, 4))
    mean, var = engine.predict(Xs, include_noise=False)
    np.testing.assert_allclose(mean, orc.cov(parts, theta, Xs, X) @ alpha, rtol=0, atol=1e-8 * np.abs(Y).max())
    assert np.all(var >= 1e-15) and np.all(var <= theta[0] + 1e-12)
    assert np.abs(mean - cases.lf_4d(Xs)).max() < 0.05            # it also regresses the function
    # (3) permutation invariance of NLML and gradient
    perm = rng.permutation(N)
    engine.set_data(X[perm], Y[perm])
    nlml_p, grad_p = engine.eval(theta, noise)
    assert nlml_p == pytest.approx(nlml, rel=1e-10)
    tol.check_grad(grad_p, grad, label="cfg3/LF N=16384 permutation")
    # (4) the gradient against a directional central difference of the HIP objective itself
    d = rng.standard_normal(3)
    d /= np.linalg.norm(d)
    p = np.concatenate([theta, [noise]])
    h = 1e-5
    fp = engine.eval((p + h * d * p)[:-1], (p + h * d * p)[-1], want_grad=False)
    fm = engine.eval((p - h * d * p)[:-1], (p - h * d * p)[-1], want_grad=False)
    assert (fp - fm) / (2 * h) == pytest.approx(float(grad_p @ (d * p)), rel=5e-4)


def test_cfg3_two_level_flow_hf_n4096_against_oracle():
    """cfg3 end to end at full size through the model surface: data-driven LF GP on 16384 points, HF composite
    level on 4096 augmented points (D = 5), then the fitted HF level against the oracle at the fitted
    hyper-parameters (N = 4096: a few seconds of host LAPACK)."""
    import multifidelity_datafusion_gps_amd as mf
    rng = np.random.default_rng(2)
    X_lf = rng.uniform(size=(16384, 4))
    X_hf = rng.uniform(size=(4096, 4))
    Xs = rng.uniform(size=(500, 4))
    model = _budgeted(evals=4, restarts=2, conc=2)(4, col(cases.hf_4d), None, lf_X=X_lf, lf_Y=col(cases.lf_4d)(X_lf),
                                                   seed=2)
    model.fit(X_hf)
    assert model.lf_model.X.shape == (16384, 4) and model.hf_model.X.shape == (4096, 5)
    # the augmentation column is the LF posterior mean: check it against the oracle's kernel rows x the device alpha
    alpha_lf = model.lf_model._engine.get_alpha()
    th_lf = np.array([model.lf_model.kern.variance.value, model.lf_model.kern.lengthscale.value])
    sample = rng.choice(4096, 128, replace=False)
    want = orc.cov(cases.single(cases.RBF, 4), th_lf, X_hf[sample], X_lf) @ alpha_lf
    np.testing.assert_allclose(model.hf_model.X[sample, 4], want, rtol=0, atol=1e-8 * np.abs(want).max())
    mean, var = model.predict(Xs)
    _check_fitted_level(model, Xs, mean[:, 0], var[:, 0], "cfg3/HF N=4096")
    assert float(np.mean((mean - col(cases.hf_4d)(Xs)) ** 2)) < 1e-2
    model.close()


# ---------------------------------------------------------------------------------------------------------------
# cfg4
# ---------------------------------------------------------------------------------------------------------------
def f3(x):
    return (np.sin(10 * x[:, 0]) ** 2 + np.cos(10 * x[:, 1]))[:, None]


def f2(x):
    return 1.5 * f3(x) + 3


def f1(x):
    return f2(x) - 1.2 * (np.sin(0.1 * np.pi * x[:, :1]) + np.sin(0.1 * np.pi * x[:, 1:2]))


def test_cfg4_three_chained_levels_n8192():
    """Three fidelity levels at N = 8192 each (the reference's classes are two-level; level 3 takes level 2's
    posterior mean as its f_low, which is any callable: src/abstractMFGP.py:82-106).  Level 1 -> 2 is handed over on
    the device (device_chaining) and must equal the host hand-over bit for bit; the top level is checked against the
    oracle at its fitted hyper-parameters."""
    import multifidelity_datafusion_gps_amd as mf
    n = 8192
    rng = np.random.default_rng(3)
    X1, X2, X3 = (rng.uniform(size=(n, 2)) for _ in range(3))
    Xs = rng.uniform(size=(300, 2))
    level2 = {}
    for chained in (True, False):
        m = _budgeted(evals=3, restarts=1)(2, f2, None, lf_X=X1, lf_Y=f1(X1), seed=3, name="level2",
                                           device_chaining=chained)
        m.fit(X2)
        assert m._chained() == chained
        level2[chained] = (m, m.hf_model.X.copy(), m.predict(Xs))
    (m_on, aug_on, (mean_on, var_on)), (m_off, aug_off, (mean_off, var_off)) = level2[True], level2[False]
    assert np.array_equal(aug_on, aug_off)                         # same augmented training inputs
    assert [p.value for p in m_on.hf_model.parameters()] == [p.value for p in m_off.hf_model.parameters()]
    assert np.array_equal(mean_on, mean_off) and np.array_equal(var_on, var_off)
    m_off.close()
    lvl2 = m_on
    lvl3 = _budgeted(evals=3, restarts=1)(2, f3, lambda x: lvl2.predict(x)[0], seed=4, name="level3")
    lvl3.fit(X3)
    assert lvl3.hf_model.X.shape == (n, 3)
    np.testing.assert_array_equal(lvl3.hf_model.X[:, 2], lvl2.predict(X3)[0][:, 0])   # level 3 sits on level 2's mean
    mean, var = lvl3.predict(Xs)
    _check_fitted_level(lvl3, Xs, mean[:, 0], var[:, 0], "cfg4/level3 N=8192")
    assert float(np.mean((mean - f3(Xs)) ** 2)) < 0.05
    lvl2.close()
    lvl3.close()


def test_cfg4_rowblock_build_in_eight_blocks_equals_fused_eval(engine):
    """SURVEY 8(e3) at cfg4's size: K(X,X) of one N = 8192 level built as eight blocks of 1024 full rows (what eight
    ranks would build and all-gather), then factorised in place: bitwise the fused evaluation."""
    rng = np.random.default_rng(33)
    n = 8192
    X = rng.uniform(size=(n, 2))
    Xa = np.hstack([X, f1(X)])
    Y = f2(X)[:, 0]
    parts, theta, noise = cases.composite(2, 1), np.array([1.1, 0.9, 0.8, 0.3, 0.5, 0.7]), 0.01 * Y.var()
    engine.set_data(Xa, Y)
    engine.set_kernel(parts)
    f0, g0 = engine.eval(theta, noise, 1e-8)
    ptr, npad = engine.dev_matrix()
    assert npad == n and ptr
    for b in range(8):
        engine.kbuild_rows(theta, noise, 1e-8, b * 1024, (b + 1) * 1024)
    f1_, g1 = engine.eval_prebuilt(True)
    assert f1_ == f0 and np.array_equal(g1, g0)


# ---------------------------------------------------------------------------------------------------------------
# cfg5
# ---------------------------------------------------------------------------------------------------------------
def _cfg5_model(n_lf, seed, evals, restarts):
    import multifidelity_datafusion_gps_amd as mf
    rng = np.random.default_rng(4)
    X_lf = rng.uniform(size=(n_lf, 4))
    m = _budgeted(evals, restarts)(4, col(cases.hf_4d), None, lf_X=X_lf, lf_Y=col(cases.lf_4d)(X_lf), seed=seed,
                                   add_noise=True, adapt_maximizer=mf.DIRECT1Maximizer())
    return m, rng


def _against_oracle(model, rng, label, n_star=64):
    """the model's state in the add_noise regime: the stated 1e-7 tolerances against the quad-precision values, cond-derived ones
    against both predictive forms of the fp64 oracle (tests/truth.py::check_add_noise_state)"""
    Xs = rng.uniform(size=(n_star, 4))
    mean, var = model.predict(Xs)                                  # add_noise: the noise variance is 1e-6 from here on
    parts, theta, noise = _theta_noise(model)
    assert noise == 1e-6
    return truth.check_add_noise_state(label, parts, theta, noise, model.hf_model.X, model.hf_Y[:, 0], model._augment_data(Xs),
                                       model.hf_model.objective_function(), mean[:, 0], var[:, 0],
                                       jitter=model.hf_model._jitter_used)


def test_cfg5_adaptation_across_block_boundaries_rank1_append():
    """cfg5 with hyper-parameters kept (reoptimize=False): N_hf 250 -> 390 crosses the 128-row boundaries at 256 and
    384 (re-upload + refactorisation there, rank-1 appends elsewhere), add_noise=True as in the reference's
    adaptation scripts (tests/utils.py:38-47).  The final model equals the oracle's at the same hyper-parameters."""
    model, rng = _cfg5_model(2048, seed=5, evals=10, restarts=2)
    model.data_driven_lf_approach = False      # adapt the HF level only (the reference's LF adaptation is unreachable)
    model.fit(rng.uniform(size=(250, 4)))
    model.predict(rng.uniform(size=(2, 4)))    # add_noise: noise := 1e-6, one refactorisation
    evals0 = model.hf_model.n_evals
    model.adapt(140, reoptimize=False)
    assert len(model.hf_X) == 390 and model.hf_model.X.shape == (390, 5)
    # no refits: appends, plus one lazy refactorisation after each of the two boundary re-uploads
    assert model.hf_model.n_evals <= evals0 + 2
    assert len(model.acquired_points) == 140 and np.all(np.array(model.acquisition_values) <= 0)
    _against_oracle(model, rng, "cfg5/rank1_append_n390")
    model.close()


def test_cfg5_adaptation_with_refit_and_add_noise_counts_evaluations():
    """cfg5 as the reference runs it (refit after every acquisition, add_noise=True), across the boundary at 256.
    Every fit builds a fresh GPRegression, so after the last fit the evaluation count is that fit's L-BFGS-B
    evaluations plus ONE refactorisation for the noise overwrite -- however many predictions the batched DIRECT and
    the caller issue afterwards (each of them re-assigns likelihood.variance = 1e-6)."""
    model, rng = _cfg5_model(2048, seed=6, evals=5, restarts=2)
    model.data_driven_lf_approach = False
    model.fit(rng.uniform(size=(252, 4)))
    model.adapt(8)                                                  # 252 -> 260, refit each step
    assert len(model.hf_X) == 260 and model.hf_model.X.shape == (260, 5)
    fit_evals = sum(r.n_evals for r in model.hf_model.optimization_runs)
    n0 = model.hf_model.n_evals
    assert n0 == fit_evals                                          # nothing but the fit so far on this model object
    for _ in range(25):
        model.predict(rng.uniform(size=(3, 4)))
    assert model.hf_model.n_evals == n0 + 1                         # one refactorisation at noise = 1e-6, then none
    _against_oracle(model, rng, "cfg5/refit_n260")
    model.close()


def test_cfg5_candidate_panel_acquisitions_n65536():
    """cfg5 in the form BASELINE.json words it -- "predictive-variance panels" (a candidate panel of N* = 65536
    Sobol points per acquisition): every acquisition is ONE two-level predictive panel of 65536 rows (low-fidelity stencil
    means -> augmented rows -> high-fidelity variance), the new row a rank-1 append.  The panel's variances are checked
    against the oracle at the current hyper-parameters on a sample of candidates AND at the oracle's own argmax, and the
    acquired point must be the oracle's choice."""
    import time
    import multifidelity_datafusion_gps_amd as mf
    model, rng = _cfg5_model(4096, seed=8, evals=6, restarts=1)
    mx = mf.adaptation_maximizers.PanelMaximizer(n_candidates=65536, seed=11)
    model.adapt_maximizer = mx
    model.data_driven_lf_approach = False
    model.fit(rng.uniform(size=(500, 4)))
    model.predict(rng.uniform(size=(2, 4)))                 # add_noise: sigma_n^2 := 1e-6 from here on
    C = mx.candidates(model.lower_bound, model.upper_bound)
    # the panel as the acquisition sees it, against the oracle (HF level at the current hyper-parameters; the augmented rows
    # come from the model's own -- device-chained -- low-fidelity means)
    t0 = time.perf_counter()
    mean, var = model.predict(C)
    t_panel = time.perf_counter() - t0
    parts, theta, noise = _theta_noise(model)
    Ca = model._augment_data(C)
    # all 65536 rows against both forms of the fp64 oracle (cond-derived pair tolerances), the first 1024 candidates (scrambled
    # Sobol: a uniform sample) against the quad-precision values at the stated 1e-7
    st = truth.check_add_noise_state("cfg5/panel_n65536", parts, theta, noise, model.hf_model.X, model.hf_Y[:, 0], Ca,
                                     model.hf_model.objective_function(), mean[:, 0], var[:, 0],
                                     jitter=model.hf_model._jitter_used, quad_rows=1024)
    _, var_o = orc.predict_stable(parts, theta, noise, model.hf_model.X, st, Ca)
    pair = tol.fp64_pair_pred_abs(tol.cond_bound(st["K"], noise), np.abs(model.hf_Y).max())
    k_o = int(np.argmax(var_o))
    evals0 = model.hf_model.n_evals
    t0 = time.perf_counter()
    model.adapt(12, reoptimize=False)                       # 500 -> 512: the last append crosses the 128-row boundary
    t_adapt = (time.perf_counter() - t0) / 12
    pts = np.array(model.acquired_points).reshape(12, 4)
    assert var[k_o, 0] >= var[:, 0].max() - pair            # the oracle's argmax is (within the pair tolerance) the panel's maximum here
    assert np.abs(pts[0] - C[int(np.argmax(var[:, 0]))]).max() == 0.0
    assert all((np.abs(C - p).sum(axis=1) == 0).any() for p in pts) and len(np.unique(pts, axis=0)) == 12
    assert mx.last_info == {"evaluations": 65536, "panels": 1, "argmax": mx.last_info["argmax"]}
    assert model.hf_model.n_evals <= evals0 + 1 and len(model.hf_X) == 512
    _against_oracle(model, rng, "cfg5/panel_after_appends_n512")
    print("cfg5 panel form: one 65536-row two-level panel at N_hf = 500 (N_lf = 4096): %.1f ms; acquisition + append: %.1f ms"
          % (t_panel * 1e3, t_adapt * 1e3))
    model.close()


def test_cfg5_at_its_stated_size_512_to_8192(engine_cls):
    """BASELINE.json config 5 at size: N_lf = 16384 (data-driven low-fidelity GP), the high-fidelity set grown from 512 to
    8192 rows by the entropy-reduction loop (src/abstractMFGP.py:317-359) with add_noise=True (src/MFDataFusion.py:154-155:
    sigma_n^2 = 1e-6, cond(Ky) ~ 1e9-1e10) -- 7680 acquisitions as rank-1 appends at fixed hyper-parameters
    (reoptimize=False), a budgeted refit at every 1024 rows, the capacity regrowth of the device slab and the refactorisation
    at every 128-row boundary on the way.  At 1024, 2048, 4096 and 8192 rows the model is compared with the oracle at the current
    hyper-parameters (quad-precision values at the stated add_noise tolerances at 1024 and 2048 rows -- round 6: the 4096-row
    quad evaluation alone was 25 s of the suite's 300; tests/test_gpu_truth.py keeps one at that size -- and the fp64 oracle at
    cond-derived ones everywhere) and with a FRESH factorisation of the same data on a second handle: a stretch of
    up to 1023 consecutive appends must not have drifted (1e-7).  The maximiser is the batched DIRECT-L with a short
    iteration budget and the loop's diagonal prediction is cut to 8 points: the test is about the factorisation, not about
    where the points land."""
    import time
    import multifidelity_datafusion_gps_amd as mf
    t0 = time.perf_counter()
    rng = np.random.default_rng(4)
    X_lf = rng.uniform(size=(16384, 4))
    maximizer = mf.DIRECT1Maximizer()
    maximizer.maxT = 4
    Model = _budgeted(evals=3, restarts=1)
    Model.diagonal_points = 8
    model = Model(4, col(cases.hf_4d), None, lf_X=X_lf, lf_Y=col(cases.lf_4d)(X_lf), lf_hf_adapt_ratio=0, seed=5,
                  add_noise=True, adapt_maximizer=maximizer)
    model.fit(rng.uniform(size=(512, 4)))
    model.predict(rng.uniform(size=(2, 4)))                 # add_noise: sigma_n^2 := 1e-6 from here on
    fresh = engine_cls(0)
    appended = 0
    for target in (1024, 2048, 3072, 4096, 5120, 6144, 7168, 8192):
        n_before = len(model.hf_X)
        evals_before = model.hf_model.n_evals
        model.adapt(target - n_before, reoptimize=False)
        assert len(model.hf_X) == target and model.hf_model.X.shape == (target, 5)
        assert len(model.acquired_points) == target - n_before
        appended += target - n_before
        # appends only: at most one lazy refactorisation per 128-row boundary crossed (re-upload), no optimiser run
        assert model.hf_model.n_evals - evals_before <= (target - n_before) // 128 + 1
        if target in (1024, 2048, 4096, 8192):
            Xs = rng.uniform(size=(48, 4))
            mean, var = model.predict(Xs)
            parts, theta, noise = _theta_noise(model)
            assert noise == 1e-6
            Xa, Y = model.hf_model.X, model.hf_Y[:, 0]
            Xsa = model._augment_data(Xs)
            # (1) no drift: a fresh factorisation of the same rows at the same hyper-parameters
            fresh.set_data(Xa, Y); fresh.set_kernel(parts)
            nlml_fresh = fresh.eval(theta, noise, model.hf_model._jitter_used, want_grad=False)
            m_f, v_f = fresh.predict(Xsa)
            assert model.hf_model.objective_function() == pytest.approx(nlml_fresh, rel=1e-7)
            np.testing.assert_allclose(mean[:, 0], m_f, rtol=0, atol=1e-7 * max(1.0, np.abs(Y).max()))
            np.testing.assert_allclose(var[:, 0], v_f, rtol=0, atol=1e-7)
            # (2) the stated add_noise tolerances (1e-7) against the QUAD-PRECISION values up to 4096 rows -- the HIP state after
            # thousands of rank-1 appends against the true numbers -- and, at every checkpoint, both predictive forms of the fp64
            # oracle at tolerances derived from cond(Ky) (two rounded evaluations: at 8192 rows the host LAPACK run alone is 0.9e-7
            # from the HIP NLML while appended and fresh HIP factorisations agree to 1e-9): tests/truth.py
            st = truth.check_add_noise_state("cfg5_at_size/n%d" % target, parts, theta, noise, Xa, Y, Xsa,
                                             model.hf_model.objective_function(), mean[:, 0], var[:, 0],
                                             jitter=model.hf_model._jitter_used, quad_max_rows=2048)
            print("cfg5 N_hf = %d: nlml %.6f (oracle %.6f, fresh %.6f), %.1f s so far"
                  % (target, model.hf_model.objective_function(), st["nlml"], nlml_fresh, time.perf_counter() - t0))
        if target < 8192:
            model.fit(model.hf_X)                           # the budgeted refit (1 + 1 runs of 3 evaluations)
            model.predict(rng.uniform(size=(2, 4)))
    assert appended == 7680
    fresh.close()
    model.close()
    print("cfg5 at size: 512 -> 8192 in %.1f s" % (time.perf_counter() - t0))


def test_add_noise_adapt_does_not_refactorise_per_prediction():
    """add_noise=True, adapt(3) with the batched DIRECT at N ~ 1000: the high-fidelity model's
    evaluation count grows by the refits only, not by one per predict()."""
    import multifidelity_datafusion_gps_amd as mf
    rng = np.random.default_rng(12)
    model = _budgeted(evals=6, restarts=2)(4, col(cases.hf_4d), col(cases.lf_4d), seed=9, add_noise=True,
                                           adapt_maximizer=mf.DIRECT1Maximizer())
    model.fit(rng.uniform(size=(1000, 4)))
    calls = {"n": 0}
    inner = model.predict

    def counting_predict(X):
        calls["n"] += 1
        return inner(X)

    model.predict = counting_predict
    model.adapt(3)
    assert calls["n"] >= 3 * 10                                     # DIRECT issued many batched predictions per step
    # after the last refit: that fit's evaluations, nothing from the predictions that preceded it
    assert model.hf_model.n_evals == sum(r.n_evals for r in model.hf_model.optimization_runs)
    total_before = model.hf_model.n_evals
    model.adapt(2, reoptimize=False)                                # 2 acquisitions, appended
    assert model.hf_model.n_evals <= total_before + 1              # ONE refactorisation (noise -> 1e-6), never per predict
    assert len(model.hf_X) == 1005
    model.close()
