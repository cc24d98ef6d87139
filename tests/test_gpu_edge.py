"""Evaluations at the edge of the parameter domain on the HIP engine (tests/edge_points.py): wherever the CPU double -- GPy's formulas
-- returns finite numbers the engine returns the same ones; where GPy's formulas break down (0 * inf, inf - inf at a lengthscale of
1e-304 .. 5e-309) the engine returns the finite limit (K = variance * I) or a FAILED evaluation, never NaN."""
import numpy as np
import pytest

from tests import edge_points as ep
from tests import tolerances as tol
from tests.oracle_engine import OracleEngine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ep.KINDS)
def test_edge_of_the_parameter_domain(engine_cls, kind):
    ref = ep.evaluate(kind, OracleEngine())
    got = ep.evaluate(kind, None)                    # engine=None: GPRegression opens its own HIP handle
    cfs = ep.cond_factors(kind)                      # "tiny noise" leaves Ky = K + 1e-8 I: cond ~ 1e10, the factor's cap
    for name, (f, g) in got.items():
        fr, gr = ref[name]
        assert f == ep.F_FAILED or np.isfinite(f), (kind, name, f)
        assert np.all(np.isfinite(g)) or name in ("nan", "inf"), (kind, name, g)
        if name in ("nan", "inf"):
            assert f == ep.F_FAILED
            continue
        if np.isfinite(fr) and np.all(np.isfinite(gr)):
            cf, cond = cfs[name]
            assert abs(f - fr) <= tol.nlml_rel(cond) * max(1.0, abs(fr)), (kind, name, f, fr, cond)
            # optimizer-space gradient (the softplus factor is part of it); 1e-280: entries that are themselves ~1e-290
            scale = np.maximum(np.abs(gr), tol.GRAD_FLOOR * np.linalg.norm(gr))
            assert np.all(np.abs(g - gr) <= tol.GRAD_REL * cf * scale + 1e-280), (kind, name, g, gr, cf)
    # the finite limit where GPy's formulas give NaN: a vanishing lengthscale decorrelates every pair, K = variance * I
    if kind in ("rbf_ard", "matern52_ard"):
        f_tiny = got["tiny lengthscale"][0]
        assert np.isfinite(f_tiny) and f_tiny != ep.F_FAILED
