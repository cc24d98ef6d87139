"""The reference-side binding (option B) as importable code: stand-ins for the three packages the
reference's own `src/` imports around this path, each bound to this package.

    import multifidelity_datafusion_gps_amd.compat as compat
    compat.install()            # before `import src`: registers GPy, DIRECT and scipydirect in sys.modules
    import src.models as models # /root/reference/src, unchanged

  * `GPy`          -- exactly the names the reference touches (src/abstractMFGP.py:3,60,62,77-80,100-104,119-122,132-137;
                      src/MFDataFusion.py:1,93-98,155-156; src/models/GPDFC.py:26): `GPy.kern.{RBF,Matern32,Matern52}`
                      and `GPy.models.GPRegression` = the HIP-backed objects of engine.py.
  * `DIRECT`       -- `solve(objective, l, u, maxT=, algmethod=)` (src/adaptation_maximizers/DIRECT1_maximizer.py:25-26,
                      DIRECT 1.0.1's signature: objective(x, user_data) -> value, returns (x, fmin, ierror)).
  * `scipydirect`  -- `minimize(func, bounds, ...)` (scipydirect_wrapper.py:26) -> object with .x, .fun.
    Both run Gablonsky's DIRECT as shipped in scipy.optimize.direct -- the Fortran code those two packages wrap -- one
    point per callback, as the reference's callbacks expect.

`engine_factory` (tests only) replaces the engine every GPRegression creates; the product default is a HIP engine
handle (_lib.Engine) and fails loudly without the library or a GPU.  tests/golden/make_reference_l3.py runs the
reference's fit / predict / adapt through this module in the build container."""
import sys
import types

import numpy as np

from . import engine as _e
from .adaptation_maximizers.direct import gablonsky_direct


def gpy_module(engine_factory=None):
    """-> a module object that answers for `import GPy` at the reference's call sites"""
    class GPRegression(_e.GPRegression):
        def __init__(self, X, Y, kernel=None, Y_metadata=None, normalizer=None, noise_var=1.0, mean_function=None, **kw):
            if normalizer is not None or mean_function is not None or Y_metadata is not None:
                raise NotImplementedError("normalizer / mean_function / Y_metadata are not used by the reference")
            if engine_factory is not None and "engine" not in kw:
                kw["engine"] = engine_factory()
            super().__init__(X, Y, kernel=kernel, noise_var=noise_var, **kw)

    gpy = types.ModuleType("GPy")
    gpy.__doc__ = "GPy stand-in bound to multifidelity_datafusion_gps_amd.engine (only the names the reference uses)"
    gpy.kern = types.ModuleType("GPy.kern")
    gpy.kern.RBF, gpy.kern.Matern32, gpy.kern.Matern52 = _e.RBF, _e.Matern32, _e.Matern52
    gpy.kern.Kern, gpy.kern.Prod, gpy.kern.Add = _e.Kern, _e.Prod, _e.Add
    gpy.models = types.ModuleType("GPy.models")
    gpy.models.GPRegression = GPRegression
    return gpy


class _DirectResult:
    def __init__(self, x, fun, info):
        self.x, self.fun, self.success = x, fun, True
        self.message = info.get("message", "")
        self.nfev, self.nit = info.get("nf"), info.get("iterations")


def _scalar_callback(func, *extra):
    def f_batch(Xb):
        return np.array([float(np.asarray(func(np.asarray(x, dtype=np.float64), *extra)).reshape(-1)[0]) for x in np.atleast_2d(Xb)])
    return f_batch


def direct_module():
    """DIRECT 1.0.1: solve(objective, l, u, eps=1e-4, maxf=20000, maxT=6000, algmethod=0, ...) -> (x, fmin, ierror)"""
    def solve(objective, l, u, eps=1e-4, maxf=20000, maxT=6000, algmethod=0, fglobal=-1e100, fglper=0.01, volper=-1.0,
              sigmaper=-1.0, logfilename="DIRresults.txt", user_data=None):
        x, fun, _ = gablonsky_direct(_scalar_callback(objective, user_data), l, u, eps=eps, maxf=maxf, maxT=maxT,
                                     algmethod=algmethod)
        return x, fun, 0
    m = types.ModuleType("DIRECT")
    m.solve = solve
    return m


def scipydirect_module():
    """scipydirect.minimize(func, bounds, eps=1e-4, maxf=20000, maxT=6000, algmethod=0, ...) -> result(.x, .fun)"""
    def minimize(func, bounds=None, nvar=None, args=(), disp=False, eps=1e-4, maxf=20000, maxT=6000, algmethod=0,
                 fglobal=-1e100, fglper=0.01, volper=-1.0, sigmaper=-1.0, **kwargs):
        lo = np.array([b[0] for b in bounds], dtype=np.float64)
        hi = np.array([b[1] for b in bounds], dtype=np.float64)
        x, fun, info = gablonsky_direct(_scalar_callback(func, *args), lo, hi, eps=eps, maxf=maxf, maxT=maxT,
                                        algmethod=algmethod)
        return _DirectResult(x, fun, info)
    m = types.ModuleType("scipydirect")
    m.minimize = minimize
    return m


def install(engine_factory=None, force=False):
    """register the stand-ins under the names the reference imports; a real installation of a package is left alone
    unless `force`.  Returns the names that were registered."""
    done = []
    for name, make in (("GPy", lambda: gpy_module(engine_factory)), ("DIRECT", direct_module), ("scipydirect", scipydirect_module)):
        if not force:
            if name in sys.modules:
                continue
            try:
                __import__(name)
                continue
            except ImportError:
                pass
        mod = make()
        sys.modules[name] = mod
        for sub in ("kern", "models"):
            if hasattr(mod, sub):
                sys.modules["%s.%s" % (name, sub)] = getattr(mod, sub)
        done.append(name)
    return done
