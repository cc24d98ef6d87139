"""Seeded synthetic cases shared by the golden-vector generator and the parity tests.

Test functions are the closed forms of the reference's experiment scripts
(/root/reference/tests/test_mfgp_adapt_{2,3,4}d.py:9-21) -- pure numpy formulas, restated here.
"""
import numpy as np

RBF, M32, M52 = 0, 1, 2
ARD = 0x100     # OR-ed into a type: one lengthscale per active column (include/mfgp.h MFGP_KERN_ARD)


def hf_2d(x):
    return np.sin(2.2 * np.pi * x[:, 0]) * np.sin(np.pi * x[:, 1])


def lf_2d(x):
    return hf_2d(x) - 1.2 * (np.sin(x[:, 0] * np.pi * 0.1) + np.sin(x[:, 1] * np.pi * 0.1))


def hf_3d(x):
    a = [3.2 * np.pi, 2.1 * np.pi, 1.2 * np.pi]
    return np.sin(x[:, 0] * a[0]) * np.sin(x[:, 1] * a[1]) * np.sin(x[:, 2] * a[2]) + 5


def hf_4d(x):
    return np.prod(np.sin(np.pi * x[:, :4]), axis=1) + 5


def lf_4d(x):
    return hf_4d(x) - 0.25 * (np.sin(x[:, 0] * np.pi * 0.1) + np.sin(x[:, 1] * np.pi * 0.05)
                              + np.sin(x[:, 2] * 0.15 * np.pi) + np.sin(x[:, 3] * 0.2 * np.pi))


def forrester_hf(x):
    return (6 * x[:, 0] - 2) ** 2 * np.sin(12 * x[:, 0] - 4)


def forrester_lf(x):
    return 0.5 * forrester_hf(x) + 10 * (x[:, 0] - 0.5) - 5


def single(ktype, D):
    return [(ktype, 0, D, 0)]


def composite(d, c, t1=RBF, t2=RBF, t3=RBF):
    """k1(aug cols) * k2(std cols) + k3(std cols)   (src/abstractMFGP.py:73-80)"""
    return [(t1, d, d + c, 0), (t2, 0, d, 0), (t3, 0, d, 1)]


def make_case(name):
    """-> dict(parts, theta, noise, X, Y, Xs)"""
    rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
    if name == "rbf_3d_n50":
        X = rng.uniform(size=(50, 3)); Y = hf_3d(X)
        return dict(parts=single(RBF, 3), theta=[1.3, 0.35], noise=0.01 * Y.var(), X=X, Y=Y, Xs=rng.uniform(size=(16, 3)))
    if name == "rbf_1d_forrester_lf":
        X = np.linspace(0, 1, 50)[:, None]; Y = forrester_lf(X)
        return dict(parts=single(RBF, 1), theta=[20.0, 0.15], noise=1e-3, X=X, Y=Y, Xs=rng.uniform(size=(16, 1)))
    if name == "nargp_1d_forrester_hf":
        X = rng.uniform(size=(10, 1)); Y = forrester_hf(X)
        Xa = np.hstack([X, forrester_lf(X)[:, None]])
        Xs = rng.uniform(size=(16, 1)); Xsa = np.hstack([Xs, forrester_lf(Xs)[:, None]])
        return dict(parts=composite(1, 1), theta=[1.5, 4.0, 0.8, 0.3, 0.6, 0.5], noise=0.01 * Y.var(), X=Xa, Y=Y, Xs=Xsa)
    if name == "nargp_4d_n64":
        X = rng.uniform(size=(64, 4)); Y = hf_4d(X)
        Xa = np.hstack([X, lf_4d(X)[:, None]])
        Xs = rng.uniform(size=(16, 4)); Xsa = np.hstack([Xs, lf_4d(Xs)[:, None]])
        return dict(parts=composite(4, 1), theta=[1.2, 1.1, 0.9, 0.6, 0.4, 0.8], noise=0.01 * Y.var(), X=Xa, Y=Y, Xs=Xsa)
    if name == "gpdfc_2d_n40":  # n=2 derivatives, tau=1e-3 -> c = 5 augmentation columns
        d, tau = 2, 1e-3
        offs = np.array([[0, 0], [-1, 0], [0, -1], [-2, 0], [0, -2]], dtype=float)
        def aug(x):
            cols = [lf_2d(x + o * tau) for o in offs]
            return np.hstack([x, np.stack(cols, 1)])
        X = rng.uniform(size=(40, d)); Y = hf_2d(X); Xs = rng.uniform(size=(16, d))
        return dict(parts=composite(2, 5), theta=[1.0, 2.0, 1.0, 0.7, 0.5, 0.9], noise=0.01 * Y.var(), X=aug(X), Y=Y, Xs=aug(Xs))
    if name == "gpdf_2d_n40":   # single RBF over all d+c columns (src/abstractMFGP.py:59-60)
        c = make_case("gpdfc_2d_n40")
        c.update(parts=single(RBF, 7), theta=[1.4, 1.7])
        return c
    if name == "matern32_3d_n48":
        X = rng.uniform(size=(48, 3)); Y = hf_3d(X)
        return dict(parts=single(M32, 3), theta=[0.9, 0.6], noise=0.02, X=X, Y=Y, Xs=rng.uniform(size=(16, 3)))
    if name == "matern52_mixed_n48":
        X = rng.uniform(size=(48, 4)); Y = hf_4d(X)
        Xa = np.hstack([X, lf_4d(X)[:, None]])
        Xs = rng.uniform(size=(16, 4)); Xsa = np.hstack([Xs, lf_4d(Xs)[:, None]])
        return dict(parts=composite(4, 1, M52, RBF, M32), theta=[1.1, 1.3, 0.7, 0.8, 0.5, 0.9], noise=0.03, X=Xa, Y=Y, Xs=Xsa)
    if name == "rbf_addnoise_n60":  # the add_noise=True regime: sigma_n^2 = 1e-6 (src/MFDataFusion.py:154-155)
        X = rng.uniform(size=(60, 3)); Y = hf_3d(X)
        return dict(parts=single(RBF, 3), theta=[1.0, 0.25], noise=1e-6, X=X, Y=Y, Xs=rng.uniform(size=(16, 3)))
    # ARD lengthscales (round 3): theta = [variance, l_0 .. l_{k-1}] per ARD factor, [variance, l] per isotropic one
    if name == "rbf_ard_3d_n50":
        X = rng.uniform(size=(50, 3)); Y = hf_3d(X)
        return dict(parts=single(RBF | ARD, 3), theta=[1.3, 0.35, 0.6, 1.1], noise=0.01 * Y.var(), X=X, Y=Y, Xs=rng.uniform(size=(16, 3)))
    if name == "nargp_ard_4d_n64":   # the composite with "ARD weights" on the two input-space factors (src/models/NARGP.py:13)
        c = make_case("nargp_4d_n64")
        c.update(parts=composite(4, 1, RBF, RBF | ARD, RBF | ARD),
                 theta=[1.2, 1.1, 0.9, 0.6, 0.75, 0.5, 0.95, 0.4, 0.8, 0.55, 1.3, 0.7])
        return c
    if name == "matern_ard_mixed_n48":
        c = make_case("matern52_mixed_n48")
        c.update(parts=composite(4, 1, M52, RBF | ARD, M32 | ARD), theta=[1.1, 1.3, 0.7, 0.8, 0.6, 1.2, 0.9, 0.5, 0.9, 0.7, 1.4, 0.65])
        return c
    # mid-size vectors (round 4): several 128-blocks, so that the golden fixtures also cover the blocked factorisation -- stored
    # compactly (no K / L: alpha, diag(L), NLML, gradient, mean and BOTH variance forms)
    if name == "nargp_4d_n512":
        X = rng.uniform(size=(512, 4)); Y = hf_4d(X)
        Xa = np.hstack([X, lf_4d(X)[:, None]])
        Xs = rng.uniform(size=(32, 4)); Xsa = np.hstack([Xs, lf_4d(Xs)[:, None]])
        return dict(parts=composite(4, 1), theta=[1.2, 1.1, 0.9, 0.6, 0.4, 0.8], noise=0.01 * Y.var(), X=Xa, Y=Y, Xs=Xsa)
    if name == "rbf_3d_n1024":
        X = rng.uniform(size=(1024, 3)); Y = hf_3d(X)
        return dict(parts=single(RBF, 3), theta=[1.0, 0.3], noise=0.01 * Y.var(), X=X, Y=Y, Xs=rng.uniform(size=(32, 3)))
    if name == "matern52_mixed_ard_n1000":
        X = rng.uniform(size=(1000, 4)); Y = hf_4d(X)
        Xa = np.hstack([X, lf_4d(X)[:, None]])
        Xs = rng.uniform(size=(32, 4)); Xsa = np.hstack([Xs, lf_4d(Xs)[:, None]])
        return dict(parts=composite(4, 1, M52, RBF | ARD, M32), theta=[1.1, 1.3, 0.7, 0.8, 0.6, 1.2, 0.9, 0.5, 0.9], noise=0.02,
                    X=Xa, Y=Y, Xs=Xsa)
    raise KeyError(name)


MID_GOLDEN_CASES = ["nargp_4d_n512", "rbf_3d_n1024", "matern52_mixed_ard_n1000"]
GOLDEN_CASES = ["rbf_3d_n50", "rbf_1d_forrester_lf", "nargp_1d_forrester_hf", "nargp_4d_n64", "gpdfc_2d_n40",
                "gpdf_2d_n40", "matern32_3d_n48", "matern52_mixed_n48", "rbf_addnoise_n60",
                "rbf_ard_3d_n50", "nargp_ard_4d_n64", "matern_ard_mixed_n48"]
