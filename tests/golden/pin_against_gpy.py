"""Pin the oracle against the REAL GPy -- for the day GPy 1.9.9 / paramz 0.9.5 are installable.

Today this script cannot do its work: GPy is neither installed nor installable offline, the reference holds no numeric
fixture, and so everything under tests/golden/ certifies self-consistency with GPy's documented mathematics only -- the oracle is
"parity unpinned".  Where `import GPy` succeeds (a build container with network access; NEVER the GPU box: the file is listed in
.gpurunignore) it regenerates every quantity of the committed vectors from GPy itself and diffs them at the stated tolerances
(tests/tolerances.py), and checks the conventions the package took from memory ([GPy-recall] in engine.py / oracle/gp_oracle.py):

  * Ky = K + (sigma_n^2 + 1e-8) I in exact inference (the 1e-8 jitter constant);
  * the latent predictive variance clipped below at 1e-15, the noise variance added on top;
  * paramz' Logexp transform: f, its inverse and the gradient factor over the whole double range (limit value 36);
  * L-BFGS-B as paramz calls it: m = 10, factr = 1e7, pgtol = 1e-5, maxfun = maxiter = max_iters -- same optimum and evaluation
    count as engine.GPRegression.optimize on the same objective from the same start;
  * optimize_restarts: the first restart CONTINUES from the current point, the others start from N(0, 1) draws in optimizer space;
  * (round 6) paramz' policy for an evaluation that fails inside the optimiser (Model._objective_grads: LinAlgError / ZeroDivisionError /
    ValueError -> the objective DBL_MAX, the previous gradient clipped to +-1e10, `_fail_count` + 1, ten in a row allowed, reset by the next
    success) -- engine.GPRegression carries exactly these on the model (`_fail_count`, `_allowed_failures`, `_F_FAILED`, `_G_CLIP_FAILED`);
  * (round 6) GPy's jitchol: a factorisation that fails is retried with jitter mean(diag) * 1e-6 * 10^k, k = 0 .. 4 on top of the 1e-8.

Nothing added since round 5 changes a number these checks look at: the distributed / batched / sharded evaluation paths and round 6's
triangular-product kernels, rank-1 append with its O(N) alpha update and early exit of a failed factorisation compute the same
quantities (bitwise for the sharded paths, within the stated tolerances for the re-ordered sums) -- the committed vectors and the
tolerances in tests/tolerances.py are what GPy is held against here.

Exit codes: 0 = everything agrees, 1 = a difference (printed), 77 = GPy is not importable (nothing was checked).
It does not try to install anything and imports nothing from /root/reference.

usage: python tests/golden/pin_against_gpy.py [--verbose]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
SKIP = 77


def _import_gpy():
    try:
        import GPy          # noqa: F401
        import paramz       # noqa: F401
    except Exception as ex:  # noqa: BLE001 - any failure to import means: not available here
        print("pin_against_gpy: GPy / paramz are not importable here (%s: %s) -- nothing was checked; the oracle stays 'parity unpinned'."
              % (type(ex).__name__, ex))
        print("pin_against_gpy: run this script where `pip install GPy==1.9.9 paramz==0.9.5` is possible (never on the GPU box).")
        return None, None
    return GPy, paramz


def _kernel_from_parts(GPy, parts, theta):
    """parts = [(type | ARD, col_begin, col_end, term)], theta in the engine's layout (include/mfgp.h: per factor its variance, then
    its lengthscale(s)) -> (GPy kernel: sum over terms of products of factors, [factor objects in theta order])"""
    classes = {0: GPy.kern.RBF, 1: GPy.kern.Matern32, 2: GPy.kern.Matern52}
    factors, pos, terms = [], 0, {}
    for t, c0, c1, term in parts:
        ard = bool(int(t) & 0x100)
        nl = (c1 - c0) if ard else 1
        var, ls = float(theta[pos]), np.array(theta[pos + 1:pos + 1 + nl], dtype=float)
        pos += 1 + nl
        k = classes[int(t) & 0xff](input_dim=int(c1 - c0), variance=var, lengthscale=ls if ard else float(ls[0]), ARD=ard,
                                   active_dims=list(range(int(c0), int(c1))))
        factors.append(k)
        terms.setdefault(int(term), []).append(k)
    total = None
    for term in sorted(terms):
        prod = terms[term][0]
        for k in terms[term][1:]:
            prod = prod * k
        total = prod if total is None else total + prod
    return total, factors


def _natural_gradient(model, factors):
    """dNLML/dtheta in the engine's layout + the noise entry, from GPy's own gradients (paramz keeps dL/dtheta of the LOG LIKELIHOOD in
    `.gradient` of every parameter, natural units)"""
    g = []
    for k in factors:
        g += [-float(np.asarray(k.variance.gradient).reshape(-1)[0])]
        g += [-float(v) for v in np.asarray(k.lengthscale.gradient).reshape(-1)]
    g += [-float(np.asarray(model.likelihood.variance.gradient).reshape(-1)[0])]
    return np.array(g)


def check_golden(GPy, name, verbose=False):
    from tests import tolerances as tol
    g = np.load(os.path.join(HERE, name + ".npz"))
    parts = [tuple(int(v) for v in p) for p in g["parts"]]
    theta, noise = np.array(g["theta"], float), float(g["noise"])
    X, Y, Xs = g["X"], np.asarray(g["Y"], float).reshape(-1, 1), g["Xs"]
    kern, factors = _kernel_from_parts(GPy, parts, theta)
    m = GPy.models.GPRegression(X, Y, kernel=kern, noise_var=noise)
    out = []
    K = kern.K(X)
    cond = tol.cond_bound(K, noise)
    tight = noise >= 1e-4
    if "K" in g.files:
        lmin = float(min(np.min(np.asarray(k.lengthscale)) for k in factors))
        vmax = float(max(float(np.asarray(k.variance).reshape(-1)[0]) for k in factors))
        out.append(("K", np.abs(K - g["K"]).max(), 2e-13 * max(vmax, 1.0) ** 2 * (1 + 1 / lmin ** 2)))
    nlml = float(m.objective_function())
    out.append(("nlml (rel)", abs(nlml - float(g["nlml"])) / abs(float(g["nlml"])), tol.nlml_rel(cond) if tight else tol.fp64_pair_nlml_rel(cond)))
    if "logdet" in g.files:
        L = np.asarray(m.posterior.woodbury_chol)
        logdet = 2.0 * np.log(np.diag(L)).sum()
        out.append(("logdet (rel)", abs(logdet - float(g["logdet"])) / abs(float(g["logdet"])), tol.nlml_rel(cond) if tight else tol.fp64_pair_nlml_rel(cond)))
    grad = _natural_gradient(m, factors)
    scale = tol.grad_scale(g["grad"])
    out.append(("gradient (per component / scale)", (np.abs(grad - g["grad"]) / scale).max(),
                tol.GRAD_REL * tol.cond_factor(cond) if tight else tol.GRAD_REL_ADDNOISE))
    mu, var = m.predict(Xs)
    ys = max(1.0, float(np.abs(Y).max()))
    kss = float(kern.Kdiag(Xs[:1])[0])
    out.append(("mean", np.abs(mu[:, 0] - g["mean"]).max(), tol.PRED_ABS * tol.cond_factor(cond) * ys if tight else tol.fp64_pair_pred_abs(cond, ys)))
    out.append(("variance (GPy's explicit-inverse form, noise included)", np.abs(var[:, 0] - g["var"]).max(),
                tol.explicit_inverse_bound(cond, kss, ys)))
    bad = [(what, err, bound) for what, err, bound in out if not err <= bound]
    if verbose or bad:
        for what, err, bound in out:
            print("  %-28s %-56s %.3e (tolerance %.1e)%s" % (name, what, err, bound, "" if err <= bound else "   <-- DIFFERS"))
    return not bad


def check_conventions(GPy, paramz, verbose=False):
    from multifidelity_datafusion_gps_amd import engine as gp
    from tests.oracle_engine import OracleEngine
    ok = True

    def report(what, good, detail=""):
        nonlocal ok
        ok = ok and bool(good)
        if verbose or not good:
            print("  convention: %-70s %s %s" % (what, "ok" if good else "DIFFERS", detail))

    rng = np.random.default_rng(0)
    X = rng.uniform(size=(30, 2))
    Y = (np.sin(5.0 * X[:, :1]) * X[:, 1:2])
    # (1) the 1e-8 jitter constant: woodbury_chol is the factor of K + (noise + 1e-8) I
    m = GPy.models.GPRegression(X, Y, GPy.kern.RBF(2, variance=1.3, lengthscale=0.4), noise_var=0.05)
    L = np.asarray(m.posterior.woodbury_chol)
    shift = float(np.mean(np.diag(L.dot(L.T) - m.kern.K(X)))) - 0.05
    report("Ky = K + (noise + 1e-8) I", abs(shift - 1e-8) < 1e-10, "(diagonal shift beyond the noise: %.3e)" % shift)
    # (2) latent variance clipped at 1e-15, noise added on top: a training point at a tiny noise variance
    m2 = GPy.models.GPRegression(X, Y, GPy.kern.RBF(2, variance=1.0, lengthscale=2.0), noise_var=1e-10)
    _, v = m2.predict(X[:5])
    report("predictive variance = max(latent, 1e-15) + noise", np.all(v >= 1e-15 + 1e-10 - 1e-25) and np.all(v < 1e-6), "(min %.3e)" % v.min())
    # (3) Logexp over the double range
    T = paramz.transformations.Logexp()
    xs = np.concatenate([np.linspace(-745.0, 745.0, 2981), [-36.0, 36.0, -1e-300, 1e-300, 0.0]])
    with np.errstate(all="ignore"):
        f_ref = np.asarray(T.f(xs.copy()), dtype=float)
        f_own = gp._logexp_f(xs.copy())
        good = np.array_equal(f_ref, f_own)
        fs = np.exp(np.linspace(np.log(1e-300), np.log(1e300), 1201))
        inv_ref = np.asarray(T.finv(fs.copy()), dtype=float)
        inv_own = gp._logexp_finv(fs.copy())
        good = good and np.allclose(inv_ref, inv_own, rtol=1e-15, atol=0)
        df = rng.standard_normal(fs.size)
        gf_ref = np.asarray(T.gradfactor(fs.copy(), df.copy()), dtype=float)
        gf_own = gp._logexp_gradfactor(fs.copy(), df.copy())
        good = good and np.allclose(gf_ref, gf_own, rtol=1e-15, atol=0)
    report("paramz Logexp: f / finv / gradfactor over the double range", good)
    # (4) L-BFGS-B controls: the same run on the same objective from the same start
    ref = GPy.models.GPRegression(X, Y, GPy.kern.RBF(2, ARD=True))
    own = gp.GPRegression(X, Y, kernel=gp.RBF(2, ARD=True), engine=OracleEngine())
    x0 = np.array(ref.optimizer_array, dtype=float)
    report("default start in optimizer space", np.allclose(own.optimizer_array, x0, rtol=1e-15))
    ref.optimize(max_iters=40)
    run = own.optimize(max_iters=40)
    report("optimize(max_iters): same optimum (L-BFGS-B m = 10, factr = 1e7, pgtol = 1e-5, maxfun = maxiter)",
           np.allclose(ref.optimizer_array, run.x_opt, rtol=1e-6, atol=1e-8) and abs(float(ref.objective_function()) - run.f_opt) <= 1e-8 * abs(run.f_opt),
           "(GPy %.10g, here %.10g)" % (float(ref.objective_function()), run.f_opt))
    # (5) optimize_restarts: restart 0 continues from the current point, the others start from fresh N(0, 1) draws
    ref2 = GPy.models.GPRegression(X, Y, GPy.kern.RBF(2))
    ref2.optimize(max_iters=5)
    here = np.array(ref2.optimizer_array, dtype=float)
    starts = []
    orig = ref2.optimize

    def spy(*a, **kw):
        starts.append(np.array(ref2.optimizer_array, dtype=float))
        return orig(*a, **kw)
    ref2.optimize = spy
    np.random.seed(123)
    ref2.optimize_restarts(num_restarts=3, verbose=False, max_iters=3)
    np.random.seed(123)
    draws = [np.random.normal(size=here.size) for _ in range(2)]
    good = len(starts) == 3 and np.array_equal(starts[0], here)
    good = good and all(np.allclose(T.finv(T.f(d.copy())), s, rtol=1e-12) or np.allclose(d, s, rtol=1e-12) for d, s in zip(draws, starts[1:]))
    report("optimize_restarts: restart 0 continues, the others from N(0, 1) draws in optimizer space", good)
    # (6) the failure policy inside an optimiser run (paramz Model._objective_grads)
    ref3 = GPy.models.GPRegression(X, Y, GPy.kern.RBF(2))
    f_good, g_good = ref3._objective_grads(np.array(ref3.optimizer_array, dtype=float))
    allowed = int(getattr(ref3, "_allowed_failures", -1))
    orig_obj = ref3.objective_function

    def failing():
        raise np.linalg.LinAlgError("injected")
    ref3.objective_function = failing
    outs = []
    for _ in range(3):
        outs.append(ref3._objective_grads(np.array(ref3.optimizer_array, dtype=float)))
    fails_after = int(ref3._fail_count)
    ref3.objective_function = orig_obj
    ref3._objective_grads(np.array(ref3.optimizer_array, dtype=float))
    good = allowed == 10 == gp.GPRegression(X, Y, kernel=gp.RBF(2), engine=OracleEngine())._allowed_failures
    good = good and all(o[0] == np.finfo(np.float64).max == gp._F_FAILED for o in outs) and fails_after == 3 and int(ref3._fail_count) == 0
    good = good and all(np.all(np.abs(o[1]) <= gp._G_CLIP_FAILED) for o in outs) and gp._G_CLIP_FAILED == 1e10
    report("a failed evaluation: DBL_MAX, the previous gradient clipped to +-1e10, ten in a row, reset by a success", good,
           "(_allowed_failures %d, _fail_count after three failures %d)" % (allowed, fails_after))
    # (7) jitchol's retries: mean(diag) * 1e-6 * 10^k
    from GPy.util.linalg import jitchol
    A = np.ones((6, 6)) * 2.0                          # rank one: the plain factorisation fails
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Lj = jitchol(A.copy())
    added = float(np.mean(np.diag(Lj.dot(Lj.T) - A)))
    ks = [k for k in range(5) if abs(added - 2.0 * 1e-6 * 10 ** k) <= 1e-3 * added]
    report("jitchol: the first jitter mean(diag) * 1e-6 * 10^k that factorises", len(ks) == 1, "(added %.3e)" % added)
    return ok


def main(argv):
    verbose = "--verbose" in argv
    GPy, paramz = _import_gpy()
    if GPy is None:
        return SKIP
    print("pin_against_gpy: GPy %s, paramz %s" % (getattr(GPy, "__version__", "?"), getattr(paramz, "__version__", "?")))
    from tests import cases
    ok = True
    for name in cases.GOLDEN_CASES + cases.MID_GOLDEN_CASES:
        ok = check_golden(GPy, name, verbose) and ok
    ok = check_conventions(GPy, paramz, verbose) and ok
    print("pin_against_gpy: %s" % ("every committed vector and every [GPy-recall] convention agrees with GPy -- the oracle is PINNED by this run"
                                   if ok else "DIFFERENCES found (above): the oracle does not restate GPy there"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
