"""Kernel and parameter objects of the GPy slice the reference touches: Param (softplus-constrained, regexp
addressable), RBF / Matern32 / Matern52 over active dimensions, Prod / Add, the Gaussian likelihood -- the objects
`GPy.kern.RBF(...)`, `k1 * k2 + k3` (/root/reference/src/abstractMFGP.py:60,77-80) and `model[".*Gaussian_noise"]` (:132-136)
resolve to.  Split out of engine.py in round 6 (one file for three concerns); engine.py re-exports every name.

Semantics restated from GPy 1.9.9 / paramz 0.9.5 (not vendored; statements tagged [GPy-recall])."""
import re

import numpy as np

from . import _lib
from ._lib import KERN_ARD, KERN_MATERN32, KERN_MATERN52, KERN_RBF

_LIM_VAL = 36.0
_LOG_LIM_VAL = np.log(np.finfo(np.float64).max)  # paramz: _log_lim_val
CONST_JITTER = 1e-8  # GPy adds this to the diagonal in exact inference [GPy-recall]


# ------------------------------------------------------------------------------------------------
# parameters
# ------------------------------------------------------------------------------------------------
def _logexp_f(x):
    # paramz 0.9.5 transformations.Logexp.f: log1p(exp(clip(x, -log(DBL_MAX), 36))), x itself above 36; the trailing
    # "+ epsilon" is commented out upstream, so nothing is added [GPy-recall]
    x = np.asarray(x, dtype=np.float64)
    return np.where(x > _LIM_VAL, x, np.log1p(np.exp(np.clip(x, -_LOG_LIM_VAL, _LIM_VAL))))


def _logexp_finv(f):
    f = np.asarray(f, dtype=np.float64)
    return np.where(f > _LIM_VAL, f, np.log(np.expm1(np.minimum(f, _LIM_VAL))))


def _logexp_gradfactor(f, df):
    f = np.asarray(f, dtype=np.float64)
    return df * np.where(f > _LIM_VAL, 1.0, -np.expm1(-f))


class Param:
    """One positive scalar hyper-parameter (variance, lengthscale, noise variance)."""

    def __init__(self, name, value, owner=None):
        self.name = name
        self._value = float(value)
        self.fixed = False
        self.gradient = 0.0
        self._observers = []
        if owner is not None:
            self._observers.append(owner)

    @property
    def value(self):
        return self._value

    @value.setter
    def value(self, v):
        v = float(np.asarray(v).reshape(-1)[0])
        if v == self._value:
            # assigning the value a parameter already has changes nothing the factorisation depends on:
            # no notification, so no O(N^3) refactorisation (MultifidelityDataFusion.predict re-assigns
            # likelihood.variance = 1e-6 on every call with add_noise=True, src/MFDataFusion.py:154-155;
            # SURVEY 8(b) allows the lazy form)
            return
        self._value = v
        for o in self._observers:
            o._param_changed(self)

    # GPy-style handles
    def fix(self):
        self.fixed = True
        return self

    def unfix(self):
        self.fixed = False
        return self

    def constrain_positive(self):  # every parameter here already lives in the positive (Logexp) domain
        return self

    def __float__(self):
        return self._value

    def __getitem__(self, i):  # GPy params are arrays: kern.lengthscale[0]
        return np.atleast_1d(self._value)[i]

    def __repr__(self):
        return "Param(%s=%.6g%s)" % (self.name, self._value, ", fixed" if self.fixed else "")


class _ParamVector:
    """the ARD lengthscales of one kernel seen as GPy sees them: one array-valued parameter"""

    def __init__(self, params):
        self.params = list(params)

    @property
    def values(self):
        return np.array([p.value for p in self.params])

    def __getitem__(self, i):
        return self.values[i]

    def __len__(self):
        return len(self.params)

    def __iter__(self):
        return iter(self.values)

    def fix(self):
        for p in self.params:
            p.fix()
        return self

    def unfix(self):
        for p in self.params:
            p.unfix()
        return self

    def constrain_positive(self):
        return self

    def __repr__(self):
        return "ParamVector(lengthscale=%s)" % np.array2string(self.values, precision=6)


# ------------------------------------------------------------------------------------------------
# kernel specification objects
# ------------------------------------------------------------------------------------------------
class Kern:
    """Base of the kernel *spec* objects: they hold parameters and column sets; the GPU evaluates them."""

    def __mul__(self, other):
        return Prod([self, other])

    def __add__(self, other):
        return Add([self, other])

    def _terms(self):
        """-> list of terms, each a list of Stationary factors (sum of products expansion)."""
        raise NotImplementedError

    def parameters(self):
        """distinct Param objects in a stable order"""
        out = []
        for term in self._terms():
            for f in term:
                for p in [f.variance] + f.lengthscales:
                    if not any(p is q for q in out):
                        out.append(p)
        return out

    def engine_parts(self):
        """flatten to the C-ABI description: parts [(type, c0, c1, term)], and per part (variance, [lengthscale Params]):
        one lengthscale for an isotropic factor, one per active column for an ARD factor (include/mfgp.h layout)"""
        parts, plist = [], []
        for t, term in enumerate(self._terms()):
            for f in term:
                parts.append((f.ktype | (KERN_ARD if f.ARD else 0), f.col_begin, f.col_end, t))
                plist.append((f.variance, list(f.lengthscales)))
        if sum(1 + len(ls) for _, ls in plist) > _lib.MAX_THETA:
            raise NotImplementedError("kernel has more than %d parameters" % _lib.MAX_THETA)
        if len(parts) > _lib.MAX_PARTS:
            raise NotImplementedError("kernel expands to %d factors; the engine supports %d" % (len(parts), _lib.MAX_PARTS))
        return parts, plist

    def _set_owner(self, owner):
        # a kernel object is linked to one live model at a time (the reference re-uses self.kernel for
        # every refit, src/MFDataFusion.py:69,96: hyper-parameters warm-start; the previous model lets go)
        for p in self.parameters():
            p._observers = [owner]

    def Kdiag_value(self):
        return sum(np.prod([f.variance.value for f in term]) for term in self._terms())


class Stationary(Kern):
    ktype = None
    _default_name = "stationary"

    def __init__(self, input_dim, variance=1.0, lengthscale=None, ARD=False, active_dims=None, name=None):
        # ARD=True: one lengthscale per input dimension (GPy Stationary [GPy-recall]; the "ARD weights" of the reference's model
        # docstrings, src/models/NARGP.py:13 -- the reference never passes the flag, its kern_class hooks are where a user would)
        self.ARD = bool(ARD)
        self.input_dim = int(input_dim)
        if active_dims is None:
            active_dims = np.arange(self.input_dim)
        active_dims = np.asarray(active_dims, dtype=int).reshape(-1)
        if len(active_dims) != self.input_dim:
            raise ValueError("len(active_dims) must equal input_dim")
        if len(active_dims) > 1 and np.any(np.diff(active_dims) != 1):
            raise NotImplementedError("active_dims must be a contiguous, ascending column range")
        self.active_dims = active_dims
        self.col_begin = int(active_dims[0])
        self.col_end = int(active_dims[-1]) + 1
        self.name = name or self._default_name
        self.variance = Param("variance", variance)
        ls = np.ones(self.input_dim if self.ARD else 1) if lengthscale is None else np.asarray(lengthscale, dtype=np.float64).reshape(-1)
        if self.ARD and ls.size == 1:
            ls = np.full(self.input_dim, ls[0])
        if ls.size != (self.input_dim if self.ARD else 1):
            raise ValueError("lengthscale must have %d entries" % (self.input_dim if self.ARD else 1))
        self.lengthscales = [Param("lengthscale" if not self.ARD else "lengthscale[%d]" % i, v) for i, v in enumerate(ls)]
        # GPy spelling: kern.lengthscale (a 1-vector, or one entry per dimension with ARD)
        self.lengthscale = self.lengthscales[0] if not self.ARD else _ParamVector(self.lengthscales)

    def _terms(self):
        return [[self]]

    def to_dict(self):
        return {"class": "GPy.kern." + type(self).__name__, "name": self.name, "input_dim": self.input_dim,
                "active_dims": self.active_dims.tolist(), "variance": [self.variance.value],
                "lengthscale": [p.value for p in self.lengthscales], "ARD": self.ARD}


class RBF(Stationary):
    ktype = KERN_RBF
    _default_name = "rbf"


class Matern32(Stationary):
    ktype = KERN_MATERN32
    _default_name = "Mat32"


class Matern52(Stationary):
    ktype = KERN_MATERN52
    _default_name = "Mat52"


class _Combination(Kern):
    def __init__(self, parts, name):
        self.parts = list(parts)
        self.name = name

    def to_dict(self):
        return {"class": "GPy.kern." + type(self).__name__, "name": self.name,
                "parts": {i: p.to_dict() for i, p in enumerate(self.parts)}}


class Prod(_Combination):
    def __init__(self, parts, name="mul"):
        flat = []
        for p in parts:
            flat.extend(p.parts if isinstance(p, Prod) else [p])
        super().__init__(flat, name)

    def _terms(self):
        terms = [[]]
        for p in self.parts:  # distribute products over sums
            terms = [a + b for a in terms for b in p._terms()]
        return terms


class Add(_Combination):
    def __init__(self, parts, name="sum"):
        flat = []
        for p in parts:
            flat.extend(p.parts if isinstance(p, Add) else [p])
        super().__init__(flat, name)

    def _terms(self):
        out = []
        for p in self.parts:
            out.extend(p._terms())
        return out


# ------------------------------------------------------------------------------------------------
# likelihood
# ------------------------------------------------------------------------------------------------
class Gaussian:
    """Gaussian likelihood: one noise variance (GPy.likelihoods.Gaussian, name 'Gaussian_noise')."""

    def __init__(self, variance=1.0, owner=None):
        self._variance = Param("Gaussian_noise.variance", variance, owner)

    @property
    def variance(self):
        return self._variance

    @variance.setter
    def variance(self, v):  # model.likelihood.variance = 1e-6 (src/MFDataFusion.py:155)
        self._variance.value = v


class _ParamSelection:
    """result of model['regex']: forwards fix/unfix/constrain_positive to the matched parameters"""

    def __init__(self, params):
        self.params = params

    def fix(self):
        for p in self.params:
            p.fix()
        return self

    def unfix(self):
        for p in self.params:
            p.unfix()
        return self

    def constrain_positive(self):
        return self

    @property
    def values(self):
        return np.array([p.value for p in self.params])

    def __len__(self):
        return len(self.params)
