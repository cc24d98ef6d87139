"""Row h of the scope table ("drops into src/models"): this package's host layer (L3) against the REFERENCE's own L3.

tests/golden/ref_l3_*.npz were made by running /root/reference/src/{abstractMFGP,MFDataFusion}.py and src/models/*.py
themselves over multifidelity_datafusion_gps_amd.compat (tests/golden/make_reference_l3.py, build container only), with a
recording CPU double as the engine.  Here the same problems go through THIS package's classes with the same double, and the
two records must coincide call by call: identical augmented design matrices (__augment_Data, src/MFDataFusion.py:177-208),
identical sequence of hyper-parameter settings and objective evaluations (ARD recipe, src/abstractMFGP.py:131-137; paramz
restart semantics), identical predictions (add_noise rule, src/MFDataFusion.py:141-156) and identical acquisitions of the
adaptation loop (src/abstractMFGP.py:317-359).  The arithmetic below the boundary is the oracle's in both runs: these
fixtures pin the host layer, not GPy's numbers."""
import os

import numpy as np
import pytest

import multifidelity_datafusion_gps_amd.models as models
from multifidelity_datafusion_gps_amd.adaptation_maximizers import ScipyDirectMaximizer
from tests import l3_problems
from tests.l3_recording import Recorder, assert_same_calls, calls_from_arrays

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _run_ours(name, adapt_steps=0, **model_kw):
    rec = Recorder()
    Eng = rec.engine_factory()

    class _Engines(dict):          # a fresh recording engine per level name, created on first use
        def __missing__(self, key):
            self[key] = Eng()
            return self[key]

    kw = dict(engines=_Engines(), adapt_maximizer=ScipyDirectMaximizer(faithful=True))
    kw.update(model_kw)
    # the engines mapping must create handles lazily: pre-seed the two level names the model asks for
    kw["engines"]["lf"], kw["engines"]["hf"]
    res = l3_problems.build_and_run(name, models, rec, adapt_steps=adapt_steps, model_kw=kw)
    return res, rec


@pytest.mark.parametrize("name", ["gpdf_2d", "nargp_2d", "gpdfc_2d", "nargp_4d"])
def test_fit_and_predict_drive_the_engine_exactly_like_the_reference(name):
    z = np.load(os.path.join(GOLDEN, "ref_l3_%s.npz" % name))
    want = calls_from_arrays(z, "log_")
    steps = 2 if "res_adapt_hf_X" in z.files else 0
    res, rec = _run_ours(name, adapt_steps=steps)
    # the reference creates its engines in the order low-fidelity (none here: f_low is a function) -> high-fidelity; ours
    # pre-creates both level handles, so engine ids are compared through the data they were given, not by number
    got = [(op, {k: v for k, v in p.items() if k != "e"}) for op, p in rec.calls]
    want = [(op, {k: v for k, v in p.items() if k != "e"}) for op, p in want]
    assert_same_calls(got, want, name)
    np.testing.assert_array_equal(res["mean"], z["res_mean"])
    np.testing.assert_array_equal(res["var"], z["res_var"])
    np.testing.assert_array_equal(res["theta"], z["res_theta"])
    if steps:
        np.testing.assert_array_equal(res["adapt_hf_X"], z["res_adapt_hf_X"])        # the same points were acquired
        np.testing.assert_array_equal(res["adapt_mean"], z["res_adapt_mean"])
        assert res["adapt_n_predict"] == int(z["res_adapt_n_predict"])                # by the same number of callbacks
        assert res["adapt_predict_sha"] == str(z["res_adapt_predict_sha"])            # at the same candidate points


def test_augmented_design_matrices_are_bit_equal_to_the_reference_loop():
    """the batched stencil stack of this package against the reference's per-row Python loop, incl. GPDF's n = 2 stencil"""
    for name in ("gpdf_2d", "gpdfc_2d", "nargp_4d"):
        z = np.load(os.path.join(GOLDEN, "ref_l3_%s.npz" % name))
        want = [p for op, p in calls_from_arrays(z, "log_") if op == "set_data"]
        _, rec = _run_ours(name)
        got = [p for op, p in rec.calls if op == "set_data"]
        assert len(got) == 1 <= len(want)        # (fixtures with adaptation steps hold the refits' matrices as well)
        for g, w in zip(got, want):
            np.testing.assert_array_equal(g["X"], w["X"])
            np.testing.assert_array_equal(g["Y"], w["Y"])
        d = l3_problems.PROBLEMS[name]["dim"]
        assert want[0]["X"].shape[1] == d + (1 if "nargp" in name else 5)      # NARGP: one column; n = 2 backward stencil: 5


def test_data_driven_low_fidelity_level_matches_the_reference():
    """lf_X / lf_Y instead of a function (src/abstractMFGP.py:95-104): the low-fidelity GP is optimised once, its posterior
    mean augments the inputs.  With the reference's calling pattern (c points per f_low call) the record is bit-equal; the
    batched stencil stack (one call per design matrix) gives the same numbers up to the rounding of a differently shaped
    matrix product."""
    z = np.load(os.path.join(GOLDEN, "ref_l3_nargp_2d_datalf.npz"))
    want = calls_from_arrays(z, "log_")
    res, rec = _run_ours("nargp_2d_datalf", batched_augmentation=False, device_chaining=False)
    got = rec.calls
    assert_same_calls([(op, {k: v for k, v in p.items() if k not in ("e", "var")}) for op, p in got],
                      [(op, {k: v for k, v in p.items() if k not in ("e", "var")}) for op, p in want], "datalf")
    np.testing.assert_array_equal(res["mean"], z["res_mean"])
    np.testing.assert_array_equal(res["var"], z["res_var"])
    res_b, rec_b = _run_ours("nargp_2d_datalf", batched_augmentation=True, device_chaining=False)
    Xa_ref = [p["X"] for op, p in want if op == "set_data"][-1]          # the high-fidelity level's augmented matrix
    Xa_b = [p["X"] for op, p in rec_b.calls if op == "set_data"][-1]
    np.testing.assert_allclose(Xa_b, Xa_ref, rtol=0, atol=1e-10)     # (K(X*,X) alpha as one GEMM instead of c-row GEMVs)
    # last-bit differences in the inputs move seven L-BFGS-B trajectories: the optimum is reproduced, not its digits
    np.testing.assert_allclose(res_b["mean"], z["res_mean"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(res_b["var"], z["res_var"], rtol=0, atol=1e-4)


def test_compat_modules_answer_for_the_reference_imports():
    """the binding of INTEGRATION.md (B): the names the reference touches resolve, and nothing else is promised"""
    import sys
    from multifidelity_datafusion_gps_amd import compat, engine
    saved = {k: sys.modules.get(k) for k in ("GPy", "GPy.kern", "GPy.models", "DIRECT", "scipydirect")}
    try:
        done = compat.install(engine_factory=Recorder().engine_factory(), force=True)
        assert done == ["GPy", "DIRECT", "scipydirect"]
        import GPy
        import DIRECT
        import scipydirect
        assert GPy.kern.RBF is engine.RBF and issubclass(GPy.models.GPRegression, engine.GPRegression)
        k = GPy.kern.RBF(1, active_dims=[2]) * GPy.kern.RBF(2, active_dims=[0, 1]) + GPy.kern.RBF(2, active_dims=[0, 1])
        assert k.to_dict()["parts"][0]["parts"][1]["lengthscale"][0] == 1.0          # src/models/GPDFC.py:26-29
        x, f, ierr = DIRECT.solve(lambda x, _: float(((x - 0.3) ** 2).sum()), np.zeros(2), np.ones(2), maxT=30, algmethod=1)
        assert ierr == 0 and np.allclose(x, 0.3, atol=2e-2)
        r = scipydirect.minimize(lambda x: np.array([[((x - 0.7) ** 2).sum()]]), [(0, 1), (0, 1)], maxT=30)
        assert np.allclose(r.x, 0.7, atol=2e-2) and r.fun < 1e-3
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_pce_driver_rounds_match_the_reference_driver():
    """SURVEY 8(f4), the caller: tests/golden/ref_l3_gpc_nargp_2d.npz was made by the REFERENCE's own round-based driver
    (/root/reference/src/gpc/mfgp_gpc.py::MFGP_GPC, a GPy-free file, run where it lies) over the reference's NARGP and this package's
    Legendre PCE.  This package's driver (gpc/mfgp_gpc.py) over this package's NARGP must produce the same rounds: the same 2 x 5
    acquisitions (same points, same number and hash of acquisition callbacks), the same polynomial-chaos moments, costs and test
    errors after every round -- bit for bit."""
    from multifidelity_datafusion_gps_amd.gpc import LegendreGPC, MFGP_GPC
    z = np.load(os.path.join(GOLDEN, "ref_l3_gpc_nargp_2d.npz"))
    rec = Recorder()
    Eng = rec.engine_factory()

    class _Engines(dict):
        def __missing__(self, key):
            self[key] = Eng()
            return self[key]

    kw = dict(engines=_Engines(), adapt_maximizer=ScipyDirectMaximizer(faithful=True))
    kw["engines"]["lf"], kw["engines"]["hf"]
    res = l3_problems.run_gpc_rounds("nargp_2d", models, rec, MFGP_GPC, LegendreGPC, num_adapts=2, model_kw=kw)
    np.testing.assert_array_equal(res["cost_history"], z["res_cost_history"])
    np.testing.assert_array_equal(res["hf_X"], z["res_hf_X"])
    np.testing.assert_array_equal(res["mean_history"], z["res_mean_history"])
    np.testing.assert_array_equal(res["var_history"], z["res_var_history"])
    np.testing.assert_array_equal(res["mse_history"], z["res_mse_history"])
    assert res["n_predict"] == int(z["res_n_predict"]) and res["predict_sha"] == str(z["res_predict_sha"])
