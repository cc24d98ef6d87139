"""The fp64 oracle (oracle/gp_oracle.py) against the quad-precision evaluation of the same quantities (oracle/quad_truth.c): K, NLML,
every gradient component, predictive mean and both forms of the predictive variance, each within the STATED tolerance of the true
value.  A third implementation of the mathematics (after scikit-learn, tests/test_oracle_vs_sklearn.py), and the only one that is exact."""
import numpy as np
import pytest

from oracle import gp_oracle as orc
from tests import truth


@pytest.mark.parametrize("name", sorted(truth.TRUTH_CASES))
def test_oracle_is_within_the_stated_tolerances_of_the_quad_precision_values(name):
    c = truth.make(name)
    tr = truth.truth_of(c)
    st = orc.inference(c["parts"], c["theta"], c["noise"], c["X"], c["Y"])
    mu, v_exp = orc.predict(c["parts"], c["theta"], c["noise"], c["X"], st, c["Xs"], include_noise=False)
    _, v_tri = orc.predict_stable(c["parts"], c["theta"], c["noise"], c["X"], st, c["Xs"], include_noise=False)
    truth.check_against_truth("oracle_vs_quad/" + name, c, tr, nlml=st["nlml"], grad=st["grad"], mean=np.ravel(mu), var=np.ravel(v_tri),
                              K=orc.cov(c["parts"], c["theta"], c["X"]), var_explicit=np.ravel(v_exp))


def test_quad_checker_against_closed_forms():
    """N = 1 and N = 2 by hand (as tests/test_oracle.py does for the oracle): the checker itself is checked"""
    from oracle import quad_truth
    from tests import cases
    v, ell, s2 = 1.7, 0.6, 0.05
    X = np.array([[0.2], [0.9]])
    y = np.array([0.3, -1.1])
    r = quad_truth.evaluate(cases.single(cases.RBF, 1), [v, ell], s2, X[:1], y[:1], X[1:], jitter=0.0, want_K=True)
    kyy = v + s2
    assert r["nlml"] == pytest.approx(0.5 * y[0] ** 2 / kyy + 0.5 * np.log(kyy) + 0.5 * np.log(2 * np.pi), rel=1e-15)
    k01 = v * np.exp(-0.5 * (0.7 / ell) ** 2)
    assert r["mean"][0] == pytest.approx(k01 * y[0] / kyy, rel=1e-15)
    assert r["var"][0] == pytest.approx(v - k01 ** 2 / kyy, rel=1e-14)
    # gradient with respect to the noise variance at N = 1: 1/2 (1/kyy - y^2/kyy^2); with respect to the variance: the same
    g = 0.5 * (1.0 / kyy - y[0] ** 2 / kyy ** 2)
    assert r["grad"][2] == pytest.approx(g, rel=1e-15) and r["grad"][0] == pytest.approx(g, rel=1e-15) and r["grad"][1] == 0.0
    r2 = quad_truth.evaluate(cases.single(cases.M32, 1), [v, ell], s2, X, y, jitter=0.0, want_K=True)
    a = np.sqrt(3.0) * 0.7 / ell
    k = v * (1 + a) * np.exp(-a)
    Ky = np.array([[v + s2, k], [k, v + s2]])
    det = Ky[0, 0] ** 2 - k ** 2
    quad = (y[0] ** 2 * Ky[1, 1] - 2 * y[0] * y[1] * k + y[1] ** 2 * Ky[0, 0]) / det
    assert r2["nlml"] == pytest.approx(0.5 * quad + 0.5 * np.log(det) + np.log(2 * np.pi), rel=1e-14)
    assert r2["K"][0, 1] == pytest.approx(k, rel=1e-15)
    # finite differences of the quad NLML reproduce its gradient (central, h = 1e-6: 1e-9 relative)
    th = np.array([v, ell])
    for i in range(2):
        hp, hm = th.copy(), th.copy()
        hp[i] += 1e-6; hm[i] -= 1e-6
        fd = (quad_truth.evaluate(cases.single(cases.M32, 1), hp, s2, X, y, jitter=0.0, want_grad=False)["nlml"]
              - quad_truth.evaluate(cases.single(cases.M32, 1), hm, s2, X, y, jitter=0.0, want_grad=False)["nlml"]) / 2e-6
        assert r2["grad"][i] == pytest.approx(fd, rel=1e-7)
