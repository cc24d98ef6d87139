"""CPU tests of the host layer that does not need the GPU: kernel-spec flattening, stencils, DIRECT, transforms,
row sharding.  (-m "not gpu")"""
import json
import os

import numpy as np
import pytest

from multifidelity_datafusion_gps_amd import engine as gp
from multifidelity_datafusion_gps_amd.adaptation_maximizers import DIRECT1Maximizer, ScipyDirectMaximizer, direct_minimize
from multifidelity_datafusion_gps_amd.augm_iterators import BackwardAugmentation, EvenAugmentation
from multifidelity_datafusion_gps_amd.sharding import LocalComm, split_rows
from oracle import gp_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_augmentation_sequences_match_reference_fixture():
    """sequences captured by running the reference's two iterator files (GPy-free) -- SURVEY.md 2.1"""
    fx = json.load(open(os.path.join(GOLD, "augm_sequences.json")))
    assert [list(map(int, v)) for v in BackwardAugmentation(2, 2)] == fx["backward_n2_dim2"]
    assert [list(map(int, v)) for v in EvenAugmentation(2, 2)] == fx["even_n2_dim2"]
    assert [list(map(int, v)) for v in BackwardAugmentation(0, 3)] == fx["backward_n0_dim3"]
    b = BackwardAugmentation(2, 2)
    assert len(list(b)) == len(list(b)) == b.new_entries_count() == 5   # re-iterable, count = n*dim + 1
    assert EvenAugmentation(3, 2).new_entries_count() == 13            # 2*n*dim + 1


def test_nargp_kernel_flattens_to_the_abi_description():
    k = gp.RBF(1, active_dims=[2]) * gp.RBF(2, active_dims=[0, 1]) + gp.RBF(2, active_dims=[0, 1])
    parts, params = k.engine_parts()
    assert parts == [(0, 2, 3, 0), (0, 0, 2, 0), (0, 0, 2, 1)]
    assert len(k.parameters()) == 6
    d = k.to_dict()  # the access path of src/models/GPDFC.py:26-29
    assert d["parts"][1]["lengthscale"] == [1.0]
    assert d["parts"][0]["parts"][0]["active_dims"] == [2]
    assert k.Kdiag_value() == 2.0
    # products distribute over sums with SHARED parameters
    a, b, c = gp.RBF(1), gp.Matern32(1), gp.Matern52(1)
    parts2, params2 = ((a + b) * c).engine_parts()
    assert [p[0] for p in parts2] == [0, 2, 1, 2] and [p[3] for p in parts2] == [0, 0, 1, 1]
    assert params2[1][0] is params2[3][0]
    with pytest.raises(NotImplementedError):
        gp.RBF(2, active_dims=[0, 2])
    with pytest.raises(NotImplementedError):
        gp.RBF(2, ARD=True)


def test_logexp_transform_matches_oracle_restatement():
    x = np.array([-40.0, -3.0, 0.0, 2.0, 40.0])
    np.testing.assert_array_equal(gp._logexp_f(x), orc.logexp_f(x))
    f = np.array([1e-3, 0.5, 2.0, 50.0])
    np.testing.assert_array_equal(gp._logexp_finv(f), orc.logexp_finv(f))
    np.testing.assert_array_equal(gp._logexp_gradfactor(f, np.ones(4)), orc.logexp_gradfactor(f, np.ones(4)))


def test_direct_finds_global_minimum_batched():
    def branin(X):
        x, y = X[:, 0], X[:, 1]
        return (y - 5.1 / (4 * np.pi ** 2) * x ** 2 + 5 / np.pi * x - 6) ** 2 + 10 * (1 - 1 / (8 * np.pi)) * np.cos(x) + 10
    calls = []

    def f(X):
        calls.append(len(X))
        return branin(X)
    for alg in (0, 1):
        x, fx, info = direct_minimize(f, [-5, 0], [10, 15], maxf=1500, algmethod=alg)
        assert fx == pytest.approx(0.397887, abs=2e-4)
        assert info["nf"] == sum(calls[-info["iterations"] - 1:]) or info["nf"] <= 1700
    assert max(calls) > 4  # evaluations arrive in batches, not one point at a time


def test_maximizers_return_negated_variance_like_the_reference():
    centre = np.array([0.3, 0.8])

    def model_predict(X):  # (means, variances): variance peaks at `centre`
        v = np.exp(-20 * np.sum((X - centre) ** 2, axis=1))[:, None]
        return np.zeros_like(v), v
    for mx in (DIRECT1Maximizer(), ScipyDirectMaximizer(maxf=3000)):
        x, fopt = mx.maximize(model_predict, np.zeros(2), np.ones(2))
        assert np.abs(x - centre).max() < 2e-2
        assert fopt == pytest.approx(-1.0, abs=2e-2)


def test_split_rows_covers_everything_once():
    for n in (0, 1, 7, 8192, 8195):
        for size in (1, 2, 3, 8):
            spans = [split_rows(n, r, size) for r in range(size)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(size - 1))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1
    c = LocalComm()
    assert c.allgather_object(5) == [5] and c.size == 1
