"""Randomised parity soak on the GPU box: random sizes (block-count boundaries of the planner included), random kernel
structures (single / composite, RBF / Matern-3/2 / -5/2, isotropic / ARD), random hyper-parameters and noise levels; the HIP
engine against the numpy/LAPACK oracle on NLML, gradient (per component), predictive mean and variance (against BOTH of the
oracle's predictive forms: GPy's explicit inverse and the triangular one) -- and, every few cases, a rank-1 append against a
fresh factorisation.  Errors are reported in units of the stated tolerances (tests/tolerances.py) times the case's conditioning
factor: <= 1 passes.  usage: fuzz_parity.py [seconds=120] [seed=0] [nmax=3000] [truth]
With `truth` as the fourth argument both fp64 paths are held against the quad-precision values instead (oracle/quad_truth.c; keep nmax
around 700: the 113-bit evaluation costs N^3)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from multifidelity_datafusion_gps_amd._lib import Engine  # noqa: E402
from oracle import gp_oracle as orc  # noqa: E402
from tests import cases  # noqa: E402
from tests import tolerances as tol  # noqa: E402

BOUNDARY_NB = [1, 2, 3, 4, 5, 8, 12, 13, 14, 15, 23, 24, 25, 26, 47, 48, 49, 55, 56, 57]   # planner defaults change around these block counts


def random_case(rng, nmax):
    if rng.uniform() < 0.5:
        nb = int(rng.choice([b for b in BOUNDARY_NB if b * 128 <= nmax + 127]))
        N = int(np.clip(nb * 128 - rng.integers(0, 3) * rng.integers(0, 127), 1, nmax))
    else:
        N = int(rng.integers(1, nmax + 1))
    d = int(rng.integers(1, 6))
    types = [cases.RBF, cases.M32, cases.M52]
    ard = lambda: cases.ARD if rng.uniform() < 0.3 else 0   # noqa: E731
    if rng.uniform() < 0.5:
        parts = [(int(rng.choice(types)) | ard(), 0, d, 0)]
        D = d
    else:
        c = int(rng.integers(1, 4))
        parts = [(int(rng.choice(types)) | ard(), d, d + c, 0), (int(rng.choice(types)) | ard(), 0, d, 0),
                 (int(rng.choice(types)) | ard(), 0, d, 1)]
        D = d + c
    npar = orc.layout(parts)[1]
    theta = np.exp(rng.uniform(np.log(0.3), np.log(3.0), size=npar))
    X = rng.uniform(size=(N, D))
    Y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.standard_normal(N)
    lo = float(os.environ.get("FUZZ_NOISE_LO", "1e-4"))      # 1e-6 .. : the add_noise regime (cond(Ky) up to ~1e9)
    noise = float(np.exp(rng.uniform(np.log(lo), np.log(0.3)))) * max(Y.var(), 1e-3)
    Xs = rng.uniform(size=(int(rng.integers(1, 200)), D))
    return dict(N=N, D=D, parts=parts, theta=theta, noise=noise, X=X, Y=Y, Xs=Xs)


def run(seconds=120.0, seed=0, nmax=3000, max_cases=None, verbose=True):
    """-> (number of cases, list of mismatches, worst relative errors)"""
    rng = np.random.default_rng(seed)
    e, fresh = Engine(0), Engine(0)
    t0, n, worst = time.time(), 0, dict(nlml=0.0, grad=0.0, mean=0.0, var=0.0, var_inv=0.0, inv_c=0.0, append=0.0)
    bad = []
    while time.time() - t0 < seconds and (max_cases is None or n < max_cases):
        c = random_case(rng, nmax)
        st = orc.inference(c["parts"], c["theta"], c["noise"], c["X"], c["Y"])
        mu, var = orc.predict_stable(c["parts"], c["theta"], c["noise"], c["X"], st, c["Xs"])
        e.set_data(c["X"], c["Y"]); e.set_kernel(c["parts"])
        nlml, grad = e.eval(c["theta"], c["noise"], 1e-8)
        m, v = e.predict(c["Xs"])
        _, var_inv = orc.predict(c["parts"], c["theta"], c["noise"], c["X"], st, c["Xs"])      # GPy's explicit-inverse form
        ys = max(1.0, np.abs(c["Y"]).max())
        # errors in units of the stated tolerances (tests/tolerances.py): NLML rel 1e-10, gradient per component 1e-8, mean /
        # variance 1e-9 max(1, |y|) -- each times the case's conditioning factor (1 up to cond(Ky) ~ 1e7)
        cond = tol.cond_bound(st["K"], c["noise"])
        cf = tol.cond_factor(cond)
        kss = orc.cov_diag(c["parts"], c["theta"], 1)[0]
        # NLML = (N log 2pi + logdet + y^T alpha) / 2 can pass through zero for random hyper-parameters while its three terms stay
        # large: the error is held against the magnitude of the terms, not of their sum (the fixed-parameter tests of the suite
        # assert rel 1e-10 of |NLML| itself: there it is not near a crossing)
        nlml_scale = max(1.0, abs(st["nlml"]), 0.5 * (c["N"] * orc.LOG_2_PI + abs(st["logdet"]) + abs(float(c["Y"] @ st["alpha"]))))
        err = dict(nlml=abs(nlml - st["nlml"]) / nlml_scale / tol.nlml_rel(cond),
                   grad=(np.abs(grad - st["grad"]) / tol.grad_scale(st["grad"])).max() / (tol.GRAD_REL * cf),
                   mean=np.abs(m - mu).max() / (tol.PRED_ABS * ys * cf),
                   var=np.abs(v - var).max() / (tol.PRED_ABS * ys * cf),
                   # GPy's explicit-inverse form: the stated tolerance, widened by that form's own eps * cond * k** error
                   var_inv=np.abs(v - var_inv).max() / tol.explicit_inverse_bound(cond, kss, ys, tol.PRED_ABS * cf),
                   inv_c=np.abs(v - var_inv).max() / (np.finfo(float).eps * cond * kss), append=0.0)
        if n % 4 == 0 and c["N"] >= 2:          # rank-1 append of the last row against the fused evaluation of all rows
            fresh.set_data(c["X"][:-1], c["Y"][:-1]); fresh.set_kernel(c["parts"])
            fresh.eval(c["theta"], c["noise"], 1e-8, want_grad=False)
            if fresh.append_row(c["X"][-1], float(c["Y"][-1])):
                m2, v2 = fresh.predict(c["Xs"])
                err["append"] = max(np.abs(m2 - m).max(), np.abs(v2 - v).max()) / (tol.PRED_ABS * ys * cf)
        scale = float(os.environ.get("FUZZ_TOL_SCALE", "1"))
        for k in worst:
            worst[k] = max(worst[k], float(err[k]))
        if any(not (err[k] <= scale) for k in err if k != "inv_c"):     # (inv_c: |var - explicit inverse| in units of eps cond k**, reported only)
            bad.append((n, c["N"], c["D"], c["parts"], [float(x) for x in c["theta"]], c["noise"], {k: float(x) for k, x in err.items()}))
            if verbose:
                print("MISMATCH", bad[-1], flush=True)
        n += 1
        if verbose and n % 25 == 0:
            print("%d cases, %.0f s, worst so far %s" % (n, time.time() - t0, {k: "%.1e" % x for k, x in worst.items()}), flush=True)
    e.close(); fresh.close()
    if verbose:
        print("fuzz_parity: %d cases in %.0f s (seed %d, N <= %d), %d mismatches; worst error / tolerance %s"
              % (n, time.time() - t0, seed, nmax, len(bad), {k: "%.2e" % x for k, x in worst.items()}))
    return n, bad, worst


def run_truth(seconds=120.0, seed=0, nmax=700, verbose=True):
    """The same random cases, but BOTH fp64 paths against the quad-precision evaluation (oracle/quad_truth.c): errors of the HIP
    engine and of the numpy/LAPACK oracle from the true values, in units of the stated tolerances x the cond factor."""
    from oracle import quad_truth
    rng = np.random.default_rng(seed)
    e = Engine(0)
    t0, n, bad = time.time(), 0, []
    worst = {w: dict(nlml=0.0, grad=0.0, mean=0.0, var=0.0) for w in ("hip", "oracle")}
    worst["oracle"]["var_explicit_over_bound"] = 0.0
    while time.time() - t0 < seconds:
        c = random_case(rng, nmax)
        if c["N"] < 2:
            continue
        tr = quad_truth.evaluate(c["parts"], c["theta"], c["noise"], c["X"], c["Y"], c["Xs"], want_K=True)
        st = orc.inference(c["parts"], c["theta"], c["noise"], c["X"], c["Y"])
        mu, var = orc.predict_stable(c["parts"], c["theta"], c["noise"], c["X"], st, c["Xs"], include_noise=False)
        _, var_inv = orc.predict(c["parts"], c["theta"], c["noise"], c["X"], st, c["Xs"], include_noise=False)
        e.set_data(c["X"], c["Y"]); e.set_kernel(c["parts"])
        nlml, grad = e.eval(c["theta"], c["noise"], 1e-8)
        m, v = e.predict(c["Xs"], include_noise=False)
        ys = max(1.0, np.abs(c["Y"]).max())
        cond = tol.cond_bound(tr["K"], c["noise"]); cf = tol.cond_factor(cond)
        scale = max(1.0, abs(tr["nlml"]), 0.5 * (c["N"] * orc.LOG_2_PI + abs(tr["logdet"]) + abs(float(c["Y"] @ tr["alpha"]))))
        tv = np.maximum(tr["var"], 1e-15)
        for who, (f_, g_, m_, v_) in (("hip", (nlml, grad, m, v)), ("oracle", (st["nlml"], st["grad"], np.ravel(mu), np.ravel(var)))):
            err = dict(nlml=abs(f_ - tr["nlml"]) / scale / tol.nlml_rel(cond),
                       grad=(np.abs(g_ - tr["grad"]) / tol.grad_scale(tr["grad"])).max() / (tol.GRAD_REL * cf),
                       mean=np.abs(m_ - tr["mean"]).max() / (tol.PRED_ABS * ys * cf),
                       var=np.abs(np.maximum(v_, 1e-15) - tv).max() / (tol.PRED_ABS * ys * cf))
            for k, x in err.items():
                worst[who][k] = max(worst[who][k], float(x))
            if any(not (x <= 1.0) for x in err.values()):
                bad.append((who, n, c["N"], c["D"], c["parts"], [float(x) for x in c["theta"]], c["noise"], {k: float(x) for k, x in err.items()}))
                if verbose:
                    print("MISMATCH", bad[-1], flush=True)
        kss = orc.cov_diag(c["parts"], c["theta"], 1)[0]
        worst["oracle"]["var_explicit_over_bound"] = max(worst["oracle"]["var_explicit_over_bound"], float(
            np.abs(np.maximum(np.ravel(var_inv), 1e-15) - tv).max() / tol.explicit_inverse_bound(cond, kss, ys, tol.PRED_ABS * cf)))
        n += 1
        if verbose and n % 50 == 0:
            print("%d cases, %.0f s" % (n, time.time() - t0), flush=True)
    e.close()
    if verbose:
        print("fuzz_parity (against quad precision): %d cases in %.0f s (seed %d, N <= %d), %d outside the stated tolerances" % (n, time.time() - t0, seed, nmax, len(bad)))
        for who in ("hip", "oracle"):
            print("  %-6s worst error / (stated tolerance x cond factor): %s" % (who, {k: "%.2e" % x for k, x in worst[who].items()}))
    return n, bad, worst


if __name__ == "__main__":
    if len(sys.argv) > 4 and sys.argv[4] == "truth":
        _, _bad, _ = run_truth(float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
        sys.exit(1 if _bad else 0)
    _, _bad, _ = run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0,
                     int(sys.argv[3]) if len(sys.argv) > 3 else 3000)
    sys.exit(1 if _bad else 0)
