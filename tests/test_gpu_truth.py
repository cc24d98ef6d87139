"""The HIP engine (through the C-ABI) against the quad-precision evaluation of the same quantities (oracle/quad_truth.c): K, NLML,
every gradient component, predictive mean and latent variance within the STATED tolerances of the true value -- N = 200 ... 4096, every
kernel family, the add_noise regime.  tests/test_oracle_truth.py holds the fp64 oracle against the same values on the CPU."""
import numpy as np
import pytest

from tests import truth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(truth.ALL))
def test_hip_outputs_are_within_the_stated_tolerances_of_the_quad_precision_values(engine, name):
    c = truth.make(name)
    tr = truth.truth_of(c)
    engine.set_data(c["X"], c["Y"])
    engine.set_kernel(c["parts"])
    if c["want_grad"]:
        nlml, grad = engine.eval(c["theta"], c["noise"], 1e-8, want_grad=True)
    else:
        nlml, grad = engine.eval(c["theta"], c["noise"], 1e-8, want_grad=False), None
    mean, var = engine.predict(c["Xs"], want_var=True, include_noise=False)
    truth.check_against_truth("hip_vs_quad/" + name, c, tr, nlml=nlml, grad=grad, mean=mean, var=var, K=engine.get_K())
