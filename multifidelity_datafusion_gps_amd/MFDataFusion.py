"""The concrete two-level model: augment high-fidelity inputs with low-fidelity evaluations, fit a GP on
the augmented inputs, predict, adapt.  Keeps the constructor, fit / adapt / predict / get_mse surface of
/root/reference/src/MFDataFusion.py:13-208 so NARGP / GPDF / GPDFC drop into src/models unchanged; all GP
arithmetic goes to libmfgp_hip.so.

Where the reference walks the rows of X in Python (one f_low call and one list per row, :177-208), this file works
on whole stencil stacks: one f_low call per design matrix, or -- with a data-driven low-fidelity level -- one device
call that never brings the low-fidelity means to the host.
"""
import numpy as np

from . import engine as gp
from .abstractMFGP import AbstractMFGP
from .adaptation_maximizers import AbstractMaximizer, ScipyDirectMaximizer
from .augm_iterators import BackwardAugmentation
from .sharding import split_rows

# adapt(plot_mode=...): which of the reference's three drawings a mode letter stands for (src/MFDataFusion.py:127-134);
# here they only select what the loop records
_PLOT_LETTERS = {"m": "plot_means", "u": "plot_uncertainties", "e": "plot_error"}
_PLOT_MODES = (None, "m", "u", "e", "mu", "um")
NOISE_FLOOR = 1e-6   # the value add_noise=True writes into the noise variance before predicting (:154-155)


def _as_design_matrix(X, input_dim, what):
    if X.ndim != 2 or X.shape[1] != input_dim:
        raise AssertionError("%s must be an (n, %d) array, got shape %s" % (what, input_dim, X.shape))
    return X


class MultifidelityDataFusion(AbstractMFGP):
    """Regression with a scarce/precise (high-fidelity) and an abundant/imprecise (low-fidelity) source.

    Parameters are those of the reference class (src/MFDataFusion.py:56-59).  Additions, all optional:
    `seed` (seeded restart draws instead of the global numpy RNG), `comm` (a sharding communicator: restarts and
    predictive panels are split over the ranks), `engines` (reuse engine handles), `batched_augmentation` (one f_low
    call on the whole (N*c, d) stencil stack instead of N calls -- identical numbers for row-wise f_low),
    `device_chaining` (with a data-driven low-fidelity level the stencil means are handed to the next level on the
    device).
    """

    def __init__(self, name: str, input_dim: int, num_derivatives: int, tau: float, f_exact: callable,
                 lower_bound: np.ndarray = None, upper_bound: np.ndarray = None, f_low: callable = None,
                 lf_X: np.ndarray = None, lf_Y: np.ndarray = None, lf_hf_adapt_ratio: int = 1,
                 use_composite_kernel: bool = True, adapt_maximizer: AbstractMaximizer = None, eps: float = 1e-8,
                 add_noise: bool = False, seed=None, comm=None, batched_augmentation: bool = True, engines=None,
                 device_chaining: bool = True):
        # a fresh maximiser per model (the reference shares one instance created at function-definition time)
        maximizer = adapt_maximizer if adapt_maximizer is not None else ScipyDirectMaximizer()
        super().__init__(name=name, input_dim=input_dim, num_derivatives=num_derivatives,
                         tau=tau, f_exact=f_exact, lower_bound=lower_bound, upper_bound=upper_bound, f_low=f_low,
                         lf_X=lf_X, lf_Y=lf_Y, lf_hf_adapt_ratio=lf_hf_adapt_ratio,
                         use_composite_kernel=use_composite_kernel, adapt_maximizer=maximizer, eps=eps)
        self.add_noise = add_noise
        self.seed = seed
        self.batched_augmentation = batched_augmentation
        self.device_chaining = device_chaining
        if comm is not None:
            self.comm = comm
        if engines:  # {"lf": Engine, "hf": Engine, ...}: reuse device-resident level state across model objects
            self._engines.update(engines)
        # order matters: the kernel's column split needs the stencil, the low-fidelity level needs the engines
        self.augm_iterator = BackwardAugmentation(self.num_derivatives, dim=input_dim)   # (:67)
        self.initialize_kernel(use_composite_kernel)
        self.initialize_lf_level(f_low, lf_X, lf_Y)

    # ---- fit ---------------------------------------------------------------------------------------------
    def fit(self, hf_X):
        """high-fidelity GP on [hf_X | f_low stencil] against f_exact(hf_X), hyper-parameters by the ARD recipe
        (src/MFDataFusion.py:75-100).  `self.kernel` is shared by every fit, so each one warm-starts from the last."""
        self.hf_X = _as_design_matrix(hf_X, self.input_dim, "hf_X")
        self.hf_Y = self.f_exact(self.hf_X)
        if self.hf_Y.shape != (len(self.hf_X), 1):
            raise AssertionError("f_exact must return an (n, 1) column, got shape %s" % (self.hf_Y.shape,))
        self._fit_count += 1
        self.hf_model = gp.GPRegression(X=self._augment_data(self.hf_X), Y=self.hf_Y, kernel=self.kernel,
                                        initialize=True, engine=self._engine("hf"))
        self.ARD(self.hf_model, self.num_restarts)

    def append_hf_point(self, x_new):
        """add one high-fidelity point WITHOUT re-optimising the hyper-parameters (not in the reference, whose
        adaptation refits from scratch every step): rank-1 Cholesky append on the device."""
        x_new = np.asarray(x_new, dtype=np.float64).reshape(1, self.input_dim)
        y_new = self.f_exact(x_new)
        self.hf_X = np.vstack((self.hf_X, x_new))
        self.hf_Y = np.vstack((self.hf_Y, y_new))
        self.hf_model.append(self._augment_data(x_new), y_new)

    # ---- adapt -------------------------------------------------------------------------------------------
    def adapt(self, adapt_steps: int, plot_mode: str = None, X_test: np.ndarray = None, Y_test: np.ndarray = None,
              eps: float = 1e-8, reoptimize: bool = True):
        """acquire up to `adapt_steps` new high-fidelity points, each where the predictive variance is largest
        (src/MFDataFusion.py:102-139).  plot_mode in {None, 'm', 'u', 'e', 'mu', 'um'} as in the reference; nothing is
        drawn, 'u' / 'e' make the loop record the test error per step.  reoptimize=False keeps the hyper-parameters
        and appends (O(N^2) per step) instead of refitting."""
        if plot_mode not in _PLOT_MODES:
            raise AssertionError("unknown plot_mode %r: choose one of %s" % (plot_mode, list(_PLOT_MODES)))
        self.adapt_steps, self.eps, self.reoptimize = adapt_steps, eps, reoptimize
        self.X_test, self.Y_test = X_test, Y_test
        if self.data_driven_lf_approach:
            self.adapt_lf()
        flags = {_PLOT_LETTERS[letter]: True for letter in (plot_mode or "")}
        self.adapt_and_plot(**flags)

    # ---- predict -----------------------------------------------------------------------------------------
    def predict(self, X_test):
        """-> (mean (N*,1), variance incl. noise (N*,1))   (src/MFDataFusion.py:141-156).
        add_noise=True overwrites the learned noise variance with 1e-6 first, as the reference does on every call
        (:154-155); re-assigning the same value later is free (engine.Param does not notify on an unchanged value)."""
        X_test = _as_design_matrix(X_test, self.input_dim, "X_test")
        if self.add_noise:
            self.hf_model.likelihood.variance = NOISE_FLOOR
        rank, size = self.comm.rank, self.comm.size
        if size == 1 or len(X_test) < 4 * size:
            return self._predict_rows(X_test)
        # predictive panels shard by rows of X*: every rank holds the replicated level state
        first, last = split_rows(len(X_test), rank, size)
        mean, var = self._predict_rows(X_test[first:last])
        both = self.comm.allgather_rows(np.hstack([mean, var]))
        return both[:, :1].copy(), both[:, 1:].copy()

    def _chained(self):
        """device-resident level chaining applies when the low-fidelity level is a GP of this package on the same
        device as the high-fidelity level"""
        return (self.device_chaining and self.data_driven_lf_approach and self.batched_augmentation
                and getattr(self.lf_model, "_engine", None) is not None and hasattr(self.lf_model._engine, "augment"))

    def _stencil(self):
        """(c, d) offsets of the low-fidelity evaluation points around x, already scaled by tau"""
        return self.augm_iterator.offsets() * self.tau

    def _predict_rows(self, X):
        hf_eng = self.hf_model._engine
        if self._chained() and hasattr(hf_eng, "predict_chained") and hf_eng.device == self.lf_model._engine.device:
            return self.hf_model.predict_chained(self.lf_model, np.ascontiguousarray(X, dtype=np.float64), self._stencil())
        return self.hf_model.predict(self._augment_data(X))

    def get_mse(self, X_test, Y_test):
        """mean squared error of the posterior mean on a test set (src/MFDataFusion.py:158-175)"""
        _as_design_matrix(X_test, self.input_dim, "X_test")
        Y_test = np.asarray(Y_test)
        if Y_test.shape != (len(X_test), 1):
            raise AssertionError("Y_test must be an (%d, 1) column of scalar targets, got shape %s"
                                 % (len(X_test), Y_test.shape))
        residual = Y_test - self.predict(X_test)[0]
        return float(np.mean(residual * residual))

    # ---- augmentation --------------------------------------------------------------------------------------
    def _augment_data(self, X):
        """[X | f_low(x + o_1 tau), ..., f_low(x + o_c tau)]  ->  (N, d + c), o_j the stencil of self.augm_iterator
        (src/MFDataFusion.py:177-208: there one f_low call per row; here one call -- or one device call -- per matrix)."""
        X = _as_design_matrix(X, self.input_dim, "X")
        stencil = self._stencil()
        if self._chained():   # stencil, low-fidelity means and the concatenation in one device call
            return self.lf_model.augment(np.ascontiguousarray(X, dtype=np.float64), stencil)
        n, c = len(X), len(stencil)
        points = X[:, None, :] + stencil[None, :, :]                  # (n, c, d): row-major = the reference's order
        if self.batched_augmentation:
            values = np.asarray(self.f_low(points.reshape(n * c, self.input_dim))).reshape(n, c)
        else:                                                          # the reference's calling pattern: c points per call
            values = np.stack([np.asarray(self.f_low(block)).reshape(c) for block in points]) if n else np.empty((0, c))
        return np.concatenate([X, values], axis=1)
