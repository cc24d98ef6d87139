"""The concrete two-level model: augment high-fidelity inputs with low-fidelity evaluations, fit a GP on
the augmented inputs, predict, adapt.  Keeps the constructor, fit / adapt / predict / get_mse surface of
/root/reference/src/MFDataFusion.py:13-208 so NARGP / GPDF / GPDFC drop into src/models unchanged; all GP
arithmetic goes to libmfgp_hip.so.
"""
import numpy as np

from . import engine as gp
from .abstractMFGP import AbstractMFGP
from .adaptation_maximizers import AbstractMaximizer, ScipyDirectMaximizer
from .augm_iterators import BackwardAugmentation
from .sharding import split_rows


class MultifidelityDataFusion(AbstractMFGP):
    """Regression with a scarce/precise (high-fidelity) and an abundant/imprecise (low-fidelity) source.

    Parameters are those of the reference class (src/MFDataFusion.py:56-59).  Additions, all optional:
    `seed` (seeded restart draws instead of the global numpy RNG), `comm` (a sharding.Comm: restarts and
    predictive panels are split over the ranks), `engines` (reuse engine handles), `batched_augmentation` (one f_low call on the whole
    (N*c, d) stencil stack instead of N calls -- identical numbers for row-wise f_low), `device_chaining` (with a
    data-driven low-fidelity level the stencil means are handed to the next level on the device).
    """

    def __init__(self, name: str, input_dim: int, num_derivatives: int, tau: float, f_exact: callable,
                 lower_bound: np.ndarray = None, upper_bound: np.ndarray = None, f_low: callable = None,
                 lf_X: np.ndarray = None, lf_Y: np.ndarray = None, lf_hf_adapt_ratio: int = 1,
                 use_composite_kernel: bool = True, adapt_maximizer: AbstractMaximizer = None, eps: float = 1e-8,
                 add_noise: bool = False, seed=None, comm=None, batched_augmentation: bool = True, engines=None,
                 device_chaining: bool = True):
        self.device_chaining = device_chaining   # before super().__init__: the LF level is built there
        if adapt_maximizer is None:  # a fresh instance per model (the reference shares one def-time instance)
            adapt_maximizer = ScipyDirectMaximizer()
        super().__init__(name=name, input_dim=input_dim, num_derivatives=num_derivatives,
                         tau=tau, f_exact=f_exact, lower_bound=lower_bound, upper_bound=upper_bound, f_low=f_low,
                         lf_X=lf_X, lf_Y=lf_Y, lf_hf_adapt_ratio=lf_hf_adapt_ratio,
                         use_composite_kernel=use_composite_kernel, adapt_maximizer=adapt_maximizer, eps=eps)
        if comm is not None:
            self.comm = comm
        if engines:  # {"lf": Engine, "hf": Engine}: reuse device-resident level state across model objects
            self._engines.update(engines)
        self.seed = seed
        self.batched_augmentation = batched_augmentation
        # augmentation stencil (src/MFDataFusion.py:67)
        self.augm_iterator = BackwardAugmentation(self.num_derivatives, dim=input_dim)
        # the kernel object is built ONCE and reused by every fit: hyper-parameters warm-start (:69, :96)
        self.initialize_kernel(use_composite_kernel)
        self.initialize_lf_level(f_low, lf_X, lf_Y)
        self.add_noise = add_noise

    def fit(self, hf_X):
        """fit the high-fidelity GP on [hf_X | f_low stencil] against f_exact(hf_X)   (src/MFDataFusion.py:75-100)"""
        assert hf_X.ndim == 2, "invalid input shape"
        assert hf_X.shape[1] == self.input_dim, "invalid input dim"
        self.hf_X = hf_X
        self.hf_Y = self.f_exact(self.hf_X)
        assert self.hf_Y.shape == (self.hf_X.shape[0], 1)
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

    def adapt(self, adapt_steps: int, plot_mode: str = None, X_test: np.ndarray = None, Y_test: np.ndarray = None,
              eps: float = 1e-8, reoptimize: bool = True):
        """acquire `adapt_steps` new high-fidelity points, each where the predictive variance is largest,
        refitting after every acquisition (src/MFDataFusion.py:102-139)."""
        self.adapt_steps = adapt_steps
        self.reoptimize = reoptimize   # False: keep the hyper-parameters and append (O(N^2) per step)
        self.X_test = X_test
        self.Y_test = Y_test
        self.eps = eps
        if self.data_driven_lf_approach:
            self.adapt_lf()
        adapt_mode_dict = {
            'u': lambda: self.adapt_and_plot(plot_uncertainties=True),
            'm': lambda: self.adapt_and_plot(plot_means=True),
            'e': lambda: self.adapt_and_plot(plot_error=True),
            'um': lambda: self.adapt_and_plot(plot_means=True, plot_uncertainties=True),
            'mu': lambda: self.adapt_and_plot(plot_means=True, plot_uncertainties=True),
            None: lambda: self.adapt_and_plot(),
        }
        assert plot_mode in adapt_mode_dict.keys(), \
            "Invalid plot mode. Select one of these: {}".format(list(adapt_mode_dict.keys()))
        adapt_mode_dict.get(plot_mode)()

    def predict(self, X_test):
        """-> (mean (N*,1), variance incl. noise (N*,1))   (src/MFDataFusion.py:141-156).
        With add_noise the learned noise is overwritten by 1e-6 before predicting (:154-155)."""
        assert X_test.ndim == 2
        assert X_test.shape[1] == self.input_dim
        if self.add_noise:
            self.hf_model.likelihood.variance = 1e-6
        size = self.comm.size
        if size > 1 and len(X_test) >= 4 * size:
            # predictive panels shard by rows of X*: every rank holds the replicated level state
            b, e = split_rows(len(X_test), self.comm.rank, size)
            m, v = self._predict_rows(X_test[b:e])
            mv = self.comm.allgather_rows(np.hstack([m, v]))
            return mv[:, :1].copy(), mv[:, 1:].copy()
        return self._predict_rows(X_test)

    def _chained(self):
        """device-resident level chaining applies when the low-fidelity level is a GP of this package on the same
        device as the high-fidelity level"""
        return (self.device_chaining and self.data_driven_lf_approach and self.batched_augmentation
                and getattr(self.lf_model, "_engine", None) is not None and hasattr(self.lf_model._engine, "augment"))

    def _predict_rows(self, X):
        if self._chained() and hasattr(self.hf_model._engine, "predict_chained") \
                and self.hf_model._engine.device == self.lf_model._engine.device:
            X = np.ascontiguousarray(X, dtype=np.float64)
            return self.hf_model.predict_chained(self.lf_model, X, self.augm_iterator.offsets() * self.tau)
        return self.hf_model.predict(self._augment_data(X))

    def get_mse(self, X_test, Y_test):
        assert len(X_test) == len(Y_test), 'unequal number of X and y values'
        assert X_test.shape[1] == self.input_dim, 'wrong input value dimension'
        assert Y_test.shape[1] == 1, 'target values must be scalars'
        preds, _ = self.predict(X_test)
        return float(np.mean((np.asarray(Y_test) - preds) ** 2))

    def _augment_data(self, X):
        """[X | f_low(x), f_low(x - tau e_0), ..., f_low(x - 2 tau e_0), ...]  ->  (N, d + c)
        (src/MFDataFusion.py:177-208; there one f_low call per row, here one call for the whole stack)."""
        assert X.shape == (len(X), self.input_dim)
        offs = self.augm_iterator.offsets()                       # (c, d)
        c = self.augm_iterator.new_entries_count()
        if self._chained():   # one device call: stencil, low-fidelity means and the concatenation
            return self.lf_model.augment(np.ascontiguousarray(X, dtype=np.float64), offs * self.tau)
        locs = X[:, None, :] + offs[None, :, :] * self.tau        # (N, c, d)
        if self.batched_augmentation:
            vals = np.asarray(self.f_low(locs.reshape(-1, self.input_dim))).reshape(len(X), c)
        else:
            vals = np.array([np.asarray(self.f_low(block)).reshape(c) for block in locs])
        augmented_X = np.concatenate([X, vals], axis=1)
        assert augmented_X.shape == (len(X), c + self.input_dim)
        return augmented_X
