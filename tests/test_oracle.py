"""CPU tests that pin the oracle (oracle/gp_oracle.py) as far as it can be pinned without GPy:
golden vectors, finite-difference gradients, closed forms, invariances.  (-m "not gpu")"""
import os

import numpy as np
import pytest

from oracle import gp_oracle as orc
from tests import cases

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", cases.GOLDEN_CASES)
def test_oracle_reproduces_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    c = cases.make_case(name)
    assert np.array_equal(c["X"], g["X"]) and np.array_equal(c["Y"], g["Y"])  # seeded inputs are stable
    parts = [tuple(p) for p in g["parts"]]
    st = orc.inference(parts, g["theta"], float(g["noise"]), g["X"], g["Y"])
    mu, var = orc.predict(parts, g["theta"], float(g["noise"]), g["X"], st, g["Xs"])
    np.testing.assert_allclose(st["K"], g["K"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(st["nlml"], g["nlml"], rtol=1e-12)
    np.testing.assert_allclose(st["grad"], g["grad"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(mu, g["mean"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(var, g["var"], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("name", cases.MID_GOLDEN_CASES)
def test_oracle_reproduces_mid_size_golden(name):
    """N = 512 .. 1024 vectors (compact: no K / L), both variance forms"""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    c = cases.make_case(name)
    assert np.array_equal(c["X"], g["X"]) and np.array_equal(c["Y"], g["Y"]) and np.array_equal(c["Xs"], g["Xs"])
    parts = [tuple(p) for p in g["parts"]]
    st = orc.inference(parts, g["theta"], float(g["noise"]), g["X"], g["Y"])
    mu, var = orc.predict(parts, g["theta"], float(g["noise"]), g["X"], st, g["Xs"])
    _, var_s = orc.predict_stable(parts, g["theta"], float(g["noise"]), g["X"], st, g["Xs"])
    np.testing.assert_allclose(st["nlml"], g["nlml"], rtol=1e-12)
    np.testing.assert_allclose(st["grad"], g["grad"], rtol=1e-8, atol=1e-9 * np.abs(g["grad"]).max())
    np.testing.assert_allclose(st["alpha"], g["alpha"], rtol=0, atol=1e-8 * np.abs(g["alpha"]).max())
    np.testing.assert_allclose(np.diag(st["L"]), g["diagL"], rtol=1e-10)
    np.testing.assert_allclose(mu, g["mean"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(var, g["var"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(var_s, g["var_stable"], rtol=0, atol=1e-11)
    assert np.abs(g["var"] - g["var_stable"]).max() <= 1e-9          # the two forms agree within the stated tolerance here


@pytest.mark.parametrize("name", cases.GOLDEN_CASES)
def test_gradient_matches_central_differences(name):
    c = cases.make_case(name)
    parts, theta, noise = c["parts"], np.array(c["theta"], float), float(c["noise"])
    st = orc.inference(parts, theta, noise, c["X"], c["Y"])
    p = np.concatenate([theta, [noise]])
    fd = np.zeros_like(p)
    for i in range(len(p)):
        h = 1e-6 * max(abs(p[i]), 1e-3)
        pp, pm = p.copy(), p.copy()
        pp[i] += h
        pm[i] -= h
        fp = orc.inference(parts, pp[:-1], pp[-1], c["X"], c["Y"], want_grad=False)["nlml"]
        fm = orc.inference(parts, pm[:-1], pm[-1], c["X"], c["Y"], want_grad=False)["nlml"]
        fd[i] = (fp - fm) / (2 * h)
    np.testing.assert_allclose(st["grad"], fd, rtol=2e-5, atol=1e-6 * np.abs(fd).max())


def test_closed_form_n1():
    X = np.array([[0.3, 0.7]])
    Y = np.array([1.7])
    for ktype in (orc.RBF, orc.MATERN32, orc.MATERN52):
        st = orc.inference([(ktype, 0, 2, 0)], [2.0, 0.5], 0.1, X, Y)
        s = 2.0 + 0.1 + 1e-8
        assert st["nlml"] == pytest.approx(0.5 * (np.log(2 * np.pi) + np.log(s) + 1.7 ** 2 / s), rel=1e-14)
        # d/d noise of the closed form
        assert st["grad"][-1] == pytest.approx(0.5 * (1 / s - 1.7 ** 2 / s ** 2), rel=1e-12)
        assert st["grad"][0] == pytest.approx(0.5 * (1 / s - 1.7 ** 2 / s ** 2), rel=1e-12)
        assert abs(st["grad"][1]) < 1e-14  # r = 0: no lengthscale dependence


def test_closed_form_kernels_n2():
    X = np.array([[0.0], [0.6]])
    r = 0.6 / 0.3
    assert orc.cov([(orc.RBF, 0, 1, 0)], [1.5, 0.3], X)[0, 1] == pytest.approx(1.5 * np.exp(-0.5 * r * r), rel=1e-15)
    assert orc.cov([(orc.MATERN32, 0, 1, 0)], [1.5, 0.3], X)[0, 1] == pytest.approx(
        1.5 * (1 + np.sqrt(3) * r) * np.exp(-np.sqrt(3) * r), rel=1e-15)
    assert orc.cov([(orc.MATERN52, 0, 1, 0)], [1.5, 0.3], X)[0, 1] == pytest.approx(
        1.5 * (1 + np.sqrt(5) * r + 5 * r * r / 3) * np.exp(-np.sqrt(5) * r), rel=1e-15)
    # composite: k1(aug)*k2(std) + k3(std) (src/abstractMFGP.py:77-80)
    Xa = np.array([[0.0, 1.0], [0.6, 3.0]])
    parts = cases.composite(1, 1)
    th = [1.2, 2.0, 0.7, 0.3, 0.4, 0.9]
    k = 1.2 * np.exp(-0.5 * (2.0 / 2.0) ** 2) * 0.7 * np.exp(-0.5 * (0.6 / 0.3) ** 2) + 0.4 * np.exp(-0.5 * (0.6 / 0.9) ** 2)
    assert orc.cov(parts, th, Xa)[0, 1] == pytest.approx(k, rel=1e-14)
    assert orc.cov_diag(parts, th, 3)[0] == pytest.approx(1.2 * 0.7 + 0.4)


def test_invariances():
    c = cases.make_case("nargp_4d_n64")
    parts, theta, noise = c["parts"], np.array(c["theta"], float), float(c["noise"])
    st = orc.inference(parts, theta, noise, c["X"], c["Y"])
    perm = np.random.default_rng(0).permutation(len(c["Y"]))
    st2 = orc.inference(parts, theta, noise, c["X"][perm], c["Y"][perm])
    assert st2["nlml"] == pytest.approx(st["nlml"], rel=1e-11)
    np.testing.assert_allclose(st2["grad"], st["grad"], rtol=1e-8, atol=1e-9)
    # scaling X and every lengthscale by the same factor leaves K unchanged
    th3 = theta.copy()
    th3[1::2] *= 3.0
    np.testing.assert_allclose(orc.cov(parts, th3, 3.0 * c["X"]), st["K"], rtol=0, atol=1e-13)


def test_predict_forms_agree_and_interpolate():
    c = cases.make_case("rbf_3d_n50")
    parts, theta = c["parts"], np.array(c["theta"], float)
    st = orc.inference(parts, theta, 1e-4, c["X"], c["Y"])
    m1, v1 = orc.predict(parts, theta, 1e-4, c["X"], st, c["Xs"])
    m2, v2 = orc.predict_stable(parts, theta, 1e-4, c["X"], st, c["Xs"])
    np.testing.assert_allclose(m1, m2, rtol=0, atol=1e-12)
    np.testing.assert_allclose(v1, v2, rtol=0, atol=1e-9)
    # at the training inputs the latent variance is below the noise level scale
    m, v = orc.predict(parts, theta, 1e-4, c["X"], st, c["X"][:5], include_noise=False)
    assert np.all(v < 2e-4)
    np.testing.assert_allclose(m, c["Y"][:5], atol=5e-2)


def test_logexp_transform_and_transformed_objective():
    x = np.array([-40.0, -3.0, 0.0, 2.0, 40.0])
    f = orc.logexp_f(x)
    assert np.all(f > 0)
    np.testing.assert_allclose(orc.logexp_finv(f)[1:], x[1:], rtol=1e-9, atol=1e-9)
    c = cases.make_case("rbf_3d_n50")
    x0 = orc.logexp_finv(np.array([1.3, 0.35, 0.05]))
    f0, g0 = orc.objective_transformed(c["parts"], x0, c["X"], c["Y"])
    for i in range(3):
        xp, xm = x0.copy(), x0.copy()
        xp[i] += 1e-6
        xm[i] -= 1e-6
        fd = (orc.objective_transformed(c["parts"], xp, c["X"], c["Y"])[0]
              - orc.objective_transformed(c["parts"], xm, c["X"], c["Y"])[0]) / 2e-6
        assert g0[i] == pytest.approx(fd, rel=1e-5, abs=1e-7)


def test_jitchol_policy():
    A = np.ones((4, 4))  # rank 1: plain dpotrf fails, jitter 1e-6*mean(diag) succeeds
    L, jit = orc.jitchol(A)
    assert jit == pytest.approx(1e-6)
    np.testing.assert_allclose(L @ L.T, A + jit * np.eye(4), atol=1e-12)
    with pytest.raises(np.linalg.LinAlgError):
        orc.jitchol(np.array([[1.0, 2.0], [2.0, -1.0]]))


def test_gpy_pinning_script_skips_cleanly_without_gpy():
    """tests/golden/pin_against_gpy.py regenerates every committed vector and the [GPy-recall] conventions from the real GPy -- where
    GPy can be imported.  Here it cannot (not installed, not installable offline): the script must say so and exit 77 without
    checking anything (and without trying to install or to import the reference); with GPy present it must pass."""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pin_against_gpy.py")
    if not os.path.exists(script):
        import pytest
        pytest.skip("the pinning script does not travel to the GPU box (.gpurunignore)")
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=600)
    try:
        import GPy  # noqa: F401
        have = True
    except Exception:  # noqa: BLE001
        have = False
    if have:
        assert r.returncode == 0, r.stdout[-3000:]
    else:
        assert r.returncode == 77 and "not importable" in r.stdout and "parity unpinned" in r.stdout, (r.returncode, r.stdout, r.stderr)
