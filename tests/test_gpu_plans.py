"""Every planner variant on the GPU against the oracle (the CPU plan checker of tests/test_plan_host.py proves the plans
race-free and exact in exact-ish arithmetic; this runs the same switches through the real kernels).  The planner reads its
switches from the environment when a handle plans, i.e. at the first mfgp_set_data of a fresh handle."""
import os

import numpy as np
import pytest

from oracle import gp_oracle as orc
from tests import cases
from tests import tolerances as tol

pytestmark = pytest.mark.gpu

VARIANTS = [{}, {"MFGP_PLAN": "levels"}, {"MFGP_PLAN": "recursive"}, {"MFGP_MACRO": "2"}, {"MFGP_MACRO": "3", "MFGP_SHIFT": "0"},
            {"MFGP_KINV_STREAM": "0"}, {"MFGP_MACRO": "4", "MFGP_SHIFT": "1"},
            {"MFGP_CHAIN_SLIM": "1", "MFGP_T128_MIN": "8"}, {"MFGP_CHAIN_SLIM": "0", "MFGP_MACRO": "1"}]
# (every switch the planner still reads -- plan.cpp's header comment; the variants rounds 1-3 measured and retired are recorded
# in tools/gemm_lab/RETIRED.md)


@pytest.mark.parametrize("env", VARIANTS, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()) or "default")
def test_planner_variant_matches_oracle(engine_cls, env):
    rng = np.random.default_rng(19)
    N = 1500                                                   # 12 leaf blocks: 3 macro panels of 4, 6 of 2
    X = rng.uniform(size=(N, 4))
    Y = cases.hf_4d(X)
    Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
    parts, theta, noise = cases.composite(4, 1), np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.01 * Y.var()
    st = orc.inference(parts, theta, noise, Xa, Y)
    Xs = rng.uniform(size=(100, 4))
    Xsa = np.hstack([Xs, cases.lf_4d(Xs)[:, None]])
    mu, var = orc.predict_stable(parts, theta, noise, Xa, st, Xsa)
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        e = engine_cls(0)
        e.set_data(Xa, Y)                                       # plans here, under the variant's switches
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    e.set_kernel(parts)
    nlml, grad = e.eval(theta, noise, 1e-8)
    tol.check_nlml(nlml, st["nlml"])
    tol.check_grad(grad, st["grad"])
    Kinv = e.get_Kinv()
    assert np.abs(Kinv - st["Kinv"]).max() <= 1e-9 * np.abs(st["Kinv"]).max()
    mean, v = e.predict(Xsa)
    tol.check_pred(mean, mu, np.abs(Y).max(), what="mean")
    tol.check_pred(v, var, np.abs(Y).max(), what="var")
    # the gradient-free factorisation + the lazy gradient (stand-alone K^-1 launch) give the same numbers
    e.factorize(theta, noise, 1e-8)
    tol.check_nlml(e.nlml(), st["nlml"])
    tol.check_grad(e.nlml_grad(), st["grad"])
    # ... and so does the batched evaluation under this variant's plan: bitwise, three points at once
    thetas = np.vstack([theta, theta * 1.1, theta * 0.9])
    fb, gb, sb = e.eval_batch(thetas, [noise, 1.5 * noise, noise], 1e-8)
    assert not sb.any() and fb[0] == nlml and np.array_equal(gb[0], grad)
    f1, g1 = e.eval(thetas[1], 1.5 * noise, 1e-8)
    assert fb[1] == f1 and np.array_equal(gb[1], g1)
    nlml2, grad2 = e.eval(theta, noise, 1e-8)
    assert nlml2 == nlml and np.array_equal(grad2, grad)        # deterministic
    e.close()


def test_beyond_the_baseline_sizes_n32768(engine_cls):
    """N = 32768 (256 leaf blocks, a 34 GB slab: four times the elements of the largest BASELINE configuration -- the absolute
    task offsets pass 2^32): the sweep plan and round 1's `levels` plan run different launches over the same arithmetic, so
    their NLML and gradient must agree to rounding, and alpha must solve the system on a sample of rows of Ky built on the
    host.  No O(N^3) host run."""
    rng = np.random.default_rng(32768)
    N = 32768
    X = rng.uniform(size=(N, 4))
    Y = cases.hf_4d(X)
    Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
    parts, theta, noise = cases.composite(4, 1), np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.01 * Y.var()
    got = []
    for env in ({}, {"MFGP_PLAN": "levels"}):
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            e = engine_cls(0)
            e.set_data(Xa, Y)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        e.set_kernel(parts)
        nlml, grad = e.eval(theta, noise, 1e-8)
        if not env:
            alpha = e.get_alpha()
            rows = rng.choice(N, size=64, replace=False)
            Krows = orc.cov(parts, theta, Xa[rows], Xa)
            Krows[np.arange(64), rows] += noise + 1e-8
            assert np.abs(Krows @ alpha - Y[rows]).max() <= 1e-10 * np.abs(Y).max()
        got.append((nlml, grad))
        e.close()
    (n0, g0), (n1, g1) = got
    assert np.isfinite(n0) and n0 == pytest.approx(n1, rel=1e-12)
    np.testing.assert_allclose(g0, g1, rtol=0, atol=1e-10 * np.abs(g1).max())


_SWITCH_DRIVER = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from multifidelity_datafusion_gps_amd._lib import Engine
from tests import cases
rng = np.random.default_rng(5)
N = 700
X = rng.uniform(size=(N, 4)); Y = cases.hf_4d(X)
Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
e = Engine(0)
e.set_data(Xa, Y); e.set_kernel(cases.composite(4, 1))
theta, noise = np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.01 * Y.var()
nlml, grad = e.eval(theta, noise, 1e-8)
Xs = rng.uniform(size=(40, 5))
out = {"nlml": nlml, "grad": grad.tolist()}
for ns in (1, 5, 20, 40):
    m, v = e.predict(Xs[:ns])
    out["m%d" % ns] = m.tolist(); out["v%d" % ns] = v.tolist()
e.close()
print(json.dumps(out))
"""


def test_process_wide_switches_in_a_process_of_their_own():
    """MFGP_SKINNY=0 (every predict through the padded tile GEMM) and MFGP_KBUILD_FAST=0 (the generic covariance / gradient kernels for
    the structures the fast paths cover) are read once per process: a child process under both switches evaluates and predicts
    (1, 5, 20 and 40 test rows: the sizes of the VALU and matrix-pipe forms) the same problem as a child under the defaults, and
    both agree with the oracle at the stated tolerances."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = {}
    for name, env in (("default", {}), ("switched", {"MFGP_SKINNY": "0", "MFGP_KBUILD_FAST": "0"})):
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", _SWITCH_DRIVER, root], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        runs[name] = json.loads(r.stdout.strip().splitlines()[-1])
    rng = np.random.default_rng(5)
    N = 700
    X = rng.uniform(size=(N, 4)); Y = cases.hf_4d(X)
    Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
    parts, theta, noise = cases.composite(4, 1), np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.01 * Y.var()
    st = orc.inference(parts, theta, noise, Xa, Y)
    Xs = rng.uniform(size=(40, 5))
    mu, var = orc.predict_stable(parts, theta, noise, Xa, st, Xs)
    for name, o in runs.items():
        tol.check_nlml(o["nlml"], st["nlml"])
        tol.check_grad(np.array(o["grad"]), st["grad"])
        for ns in (1, 5, 20, 40):
            tol.check_pred(np.array(o["m%d" % ns]), mu[:ns], np.abs(Y).max(), what="mean")
            tol.check_pred(np.array(o["v%d" % ns]), var[:ns], np.abs(Y).max(), what="var")
    for ns in (1, 5, 20, 40):     # the two routes differ in the order of their sums only
        np.testing.assert_allclose(runs["default"]["v%d" % ns], runs["switched"]["v%d" % ns], rtol=0, atol=1e-12)


_PREDV2_DRIVER = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from multifidelity_datafusion_gps_amd._lib import Engine
from tests import cases
rng = np.random.default_rng(9)
N = 7400                                    # Np = 7424 (58 leaf blocks, 116 groups of 64 rows)
X = rng.uniform(size=(N, 4)); Y = cases.hf_4d(X)
Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
e = Engine(0)
e.set_data(Xa, Y); e.set_kernel(cases.composite(4, 1))
e.factorize(np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.01 * Y.var())
Xs = rng.uniform(size=(64, 5))
out = {}
for ns in (12, 16, 20, 40, 64):
    m, v = e.predict(Xs[:ns])
    out["m%d" % ns] = m.tolist(); out["v%d" % ns] = v.tolist()
m, v = e.predict(rng.uniform(size=(200, 5))[:65] * 0 + np.vstack([Xs, Xs[:1]]))     # 65 rows: the tile GEMM, either way
out["m65"] = m.tolist(); out["v65"] = v.tolist()
e.close()
print(json.dumps(out))
"""


def test_the_two_forms_of_the_9_to_64_row_variance_product_agree():
    """MFGP_PREDV2=0 keeps every 9 .. 64-row predict on round 6's first matrix-pipe form (S through LDS-DMA, a workgroup per 16-row
    block); the default takes the register-staged form from the sizes at which it pays (shares of the triangle, partial planes).
    Read once per process: two child processes predict the same 12 / 16 / 20 / 40 / 64 rows at N = 7400; the means are bitwise
    equal (the same mean blocks ride in either launch), the variances agree to 1e-12 with each other and with the 65-row tile
    GEMM of the same rows."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = {}
    for name, env in (("default", {}), ("first_form", {"MFGP_PREDV2": "0"})):
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", _PREDV2_DRIVER, root], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        runs[name] = json.loads(r.stdout.strip().splitlines()[-1])
    a, b = runs["default"], runs["first_form"]
    assert a["m65"] == b["m65"] and a["v65"] == b["v65"]
    for ns in (12, 16, 20, 40, 64):
        assert a["m%d" % ns] == b["m%d" % ns] == a["m65"][:ns]
        np.testing.assert_allclose(a["v%d" % ns], b["v%d" % ns], rtol=0, atol=1e-12)
        np.testing.assert_allclose(a["v%d" % ns], a["v65"][:ns], rtol=0, atol=1e-12)
        assert a["v%d" % ns] != b["v%d" % ns] or ns == 0      # (different orders of the k sums: the switch did switch)
