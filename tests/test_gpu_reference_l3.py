"""The HIP engine at the states the REFERENCE's own L3 reached (tests/golden/ref_l3_*.npz, made by running
/root/reference/src over multifidelity_datafusion_gps_amd.compat: tests/golden/make_reference_l3.py).  Every predict call of
a fixture is replayed on the GPU: the design matrix the reference's __augment_Data produced, the hyper-parameters its ARD
recipe had installed at that moment (the last evaluation before the call), the test rows it asked for.  Mean and variance
must agree with what the recorded run returned (oracle arithmetic) to the tolerances of tests/test_gpu_parity.py."""
import os

import numpy as np
import pytest

from tests.l3_recording import calls_from_arrays

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["gpdf_2d", "nargp_2d", "gpdfc_2d", "nargp_4d", "nargp_2d_datalf"])
def test_hip_predictions_at_the_reference_l3_states(engine_cls, name):
    z = np.load(os.path.join(GOLDEN, "ref_l3_%s.npz" % name))
    calls = calls_from_arrays(z, "log_")
    state = {}            # engine id -> dict(X, Y, parts, theta, noise, jitter)
    engines = {}
    n_checked = 0
    for op, p in calls:
        e = int(p["e"])
        st = state.setdefault(e, {})
        if op == "set_data":
            st.update(X=p["X"], Y=p["Y"], fresh=True)
        elif op == "set_kernel":
            st.update(parts=[tuple(int(v) for v in row) for row in np.atleast_2d(p["parts"])], fresh=True)
        elif op == "eval":
            st.update(theta=p["theta"], noise=float(p["noise"]), jitter=float(p["jitter"]))
        elif op == "predict":
            eng = engines.setdefault(e, engine_cls(0))
            eng.set_data(st["X"], st["Y"])
            eng.set_kernel(st["parts"])
            nlml = eng.eval(st["theta"], st["noise"], st["jitter"], want_grad=False)
            assert np.isfinite(nlml)
            mean, var = eng.predict(p["Xs"], want_var=True, include_noise=bool(p["include_noise"]))
            # add_noise=True fixtures predict at sigma_n^2 = 1e-6 (cond ~ 1e8..1e10): the add_noise tolerances of the parity suite
            tight = st["noise"] > 1e-5
            scale = max(1.0, float(np.abs(st["Y"]).max()))
            np.testing.assert_allclose(mean, p["mean"], rtol=0, atol=(1e-9 if tight else 1e-6) * scale)
            np.testing.assert_allclose(var, p["var"], rtol=0, atol=(1e-9 if tight else 1e-6) * scale)
            n_checked += 1
    for eng in engines.values():
        eng.close()
    assert n_checked >= 1
