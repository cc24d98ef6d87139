"""Abstract multi-fidelity GP model: configuration, kernel construction, the data-driven low-fidelity
level, the hyper-parameter recipe and the entropy-reduction adaptation loop.

Fresh restatement of the orchestration surface of /root/reference/src/abstractMFGP.py (ctor :12-33,
abstract API :35-49, initialize_kernel :51-60, get_NARGP_kernel :62-80, initialize_lf_level :82-106,
adapt_lf :108-122, get_input_with_highest_uncertainty :124-129, ARD :131-137, the adaptation loop
:317-359) with every GPy call replaced by the HIP-backed objects of engine.py.  Plotting (matplotlib,
:139-273 and the drawing branches of :290-352) is out of scope; the loop records the same quantities
(`mse_history`, `acquired_points`, `acquisition_values`) instead of drawing them.
"""
import abc

import numpy as np

from . import engine as gp
from .adaptation_maximizers import AbstractMaximizer
from .sharding import LocalComm


class AbstractMFGP(metaclass=abc.ABCMeta):

    # the constants the reference hard-codes, exposed as overridable attributes (SURVEY.md section 5 "Config")
    noise_ratio = 0.01          # sigma_n^2 := 0.01 Var(Y) before the first run (src/abstractMFGP.py:132)
    first_run_max_iters = 500   # :134
    restart_max_iters = 1000    # :137
    num_restarts = 6            # src/MFDataFusion.py:100
    lf_max_iters = 1000         # lf_model.optimize() default budget (src/abstractMFGP.py:103)
    eval_cap = None             # hard cap on objective evaluations per L-BFGS-B run (benchmarks: exact budgets)
    restart_concurrency = 1     # >1: that many randomized restarts run concurrently with the first run / restart 0

    @abc.abstractmethod
    def __init__(self, name: str, input_dim: int, num_derivatives: int, tau: float, f_exact: callable,
                 lower_bound: np.ndarray, upper_bound: np.ndarray, f_low: callable, lf_X: np.ndarray, lf_Y: np.ndarray,
                 lf_hf_adapt_ratio: int, use_composite_kernel: bool, adapt_maximizer: AbstractMaximizer, eps: float):
        super().__init__()
        self.name = name
        self.input_dim = input_dim
        self.num_derivatives = num_derivatives
        self.tau = tau
        self.f_exact = f_exact
        self.f_low = f_low
        self.lf_hf_adapt_ratio = lf_hf_adapt_ratio
        self.adapt_maximizer = adapt_maximizer
        self.eps = eps
        self.comm = LocalComm()
        self.seed = None
        self._fit_count = 0
        self._engines = {}
        # data bounds: the unit box when none are given (src/abstractMFGP.py:28-33)
        if lower_bound is None and upper_bound is None:
            self.lower_bound = np.zeros(input_dim)
            self.upper_bound = np.ones(input_dim)
        else:
            self.lower_bound = lower_bound
            self.upper_bound = upper_bound

    @abc.abstractmethod
    def fit(self, hf_X):
        pass

    @abc.abstractmethod
    def adapt(self, adapt_steps, plot_mode, X_test, Y_test):
        pass

    @abc.abstractmethod
    def predict(self, X_test):
        pass

    @abc.abstractmethod
    def get_mse(self, X_test, Y_test):
        pass

    # ---- engines: one device-resident level state per fidelity level, reused by every refit ----------
    def _engine(self, level):
        if level not in self._engines:
            from ._lib import Engine
            self._engines[level] = Engine()
        return self._engines[level]

    # ---- kernels -------------------------------------------------------------------------------------
    def initialize_kernel(self, use_composite_kernel: bool):
        """composite NARGP kernel, or ONE isotropic RBF over all d + c augmented columns
        (no ARD flag is passed in the reference either, src/abstractMFGP.py:59-60)"""
        if use_composite_kernel:
            self.kernel = self.get_NARGP_kernel()
        else:
            new_input_dims = self.input_dim + self.augm_iterator.new_entries_count()
            self.kernel = gp.RBF(new_input_dims)

    def get_NARGP_kernel(self, kern_class1=gp.RBF, kern_class2=gp.RBF, kern_class3=gp.RBF):
        """k1(augmentation columns) * k2(input columns) + k3(input columns)   (src/abstractMFGP.py:73-80)"""
        std_input_dim = self.input_dim
        std_indezes = np.arange(self.input_dim)
        aug_input_dim = self.augm_iterator.new_entries_count()
        aug_indezes = np.arange(self.input_dim, self.input_dim + aug_input_dim)
        kern1 = kern_class1(aug_input_dim, active_dims=aug_indezes)
        kern2 = kern_class2(std_input_dim, active_dims=std_indezes)
        kern3 = kern_class3(std_input_dim, active_dims=std_indezes)
        return kern1 * kern2 + kern3

    # ---- low-fidelity level ----------------------------------------------------------------------------
    def initialize_lf_level(self, f_low: callable = None, lf_X: np.ndarray = None, lf_Y: np.ndarray = None):
        """exactly one of {f_low} / {lf_X, lf_Y}: a python function, or a GP trained on low-fidelity data
        whose posterior MEAN becomes f_low (src/abstractMFGP.py:93-106)."""
        lf_model_params_are_valid = (f_low is not None) ^ (
            (lf_X is not None) and (lf_Y is not None) and (self.lf_hf_adapt_ratio is not None))
        assert lf_model_params_are_valid, 'define low-fidelity model either by predicition function or by data'
        self.data_driven_lf_approach = f_low is None
        if self.data_driven_lf_approach:
            self.lf_X = lf_X
            self.lf_Y = lf_Y
            self.lf_model = gp.GPRegression(X=lf_X, Y=lf_Y, initialize=True, engine=self._engine("lf"))
            self.lf_model.eval_cap = self.eval_cap
            self.lf_model.optimize(max_iters=self.lf_max_iters)
            self.f_low = lambda t: self.lf_model.predict_mean(t)
        else:
            self.f_low = f_low

    def adapt_lf(self):
        """acquire additional low-fidelity points where the LF model is most uncertain and refit it.
        (The reference's version, src/abstractMFGP.py:108-122, is unreachable: MFDataFusion.adapt calls a
        name-mangled attribute that does not exist; this is the behaviour its docstring describes.)"""
        assert hasattr(self, 'lf_model'), "lf-model not initialized"
        for _ in range(self.adapt_steps * self.lf_hf_adapt_ratio):
            acquired_x, _ = self.adapt_maximizer.maximize(self.lf_model.predict, self.lower_bound, self.upper_bound)
            acquired_y = self.lf_model.predict(acquired_x[None])[0][0]
            self.lf_X = np.vstack((self.lf_X, acquired_x))
            self.lf_Y = np.vstack((self.lf_Y, acquired_y))
            self.lf_model = gp.GPRegression(self.lf_X, self.lf_Y, initialize=True, engine=self._engine("lf"))
            self.ARD(self.lf_model, self.num_restarts)

    def get_input_with_highest_uncertainty(self, model=None):
        """global maximiser of the model's predictive variance over the box"""
        x, fopt = self.adapt_maximizer.maximize(self.predict, self.lower_bound, self.upper_bound)
        return x, fopt

    # ---- hyper-parameter recipe ------------------------------------------------------------------------
    def _restart_rng(self):
        if self.seed is None:
            return None  # the reference draws from the global numpy RNG
        fit_id = self._fit_count
        return lambda i: np.random.default_rng([int(self.seed), fit_id, i]).normal

    def ARD(self, model, num_restarts):
        """noise := 0.01 Var(Y), fixed -> one L-BFGS-B run (500) -> free the noise -> `num_restarts`
        restarts (1000 each), best wins (src/abstractMFGP.py:131-137).  1 + num_restarts runs per fit.

        With `restart_concurrency` > 1 the randomized restarts 1.. (which, by paramz' semantics, start from fresh
        N(0,1) draws and so do not depend on the first run) execute on auxiliary engine handles in background
        threads WHILE the main thread does the first run and restart 0; same runs, same winner rule."""
        model[".*Gaussian_noise"] = model.Y.var() * self.noise_ratio
        model[".*Gaussian_noise"].fix()
        model.eval_cap = self.eval_cap
        conc = int(self.restart_concurrency)
        rank, size = self.comm.rank, self.comm.size
        if conc <= 1:
            model.optimize(max_iters=self.first_run_max_iters)
            model[".*Gaussian_noise"].unfix()
            model[".*Gaussian_noise"].constrain_positive()
            model.optimize_restarts(num_restarts, optimizer="bfgs", max_iters=self.restart_max_iters, verbose=False,
                                    rand_gen=self._restart_rng(), comm=self.comm)
            return
        # Rank 0 owns the only sequential piece (first run -> restart 0, which continues from it); the randomized
        # restarts go to the least-loaded rank.  No other rank needs the first run: the winner overwrites every
        # free parameter on every rank.
        mine_bg = self.assign_restarts(num_restarts, size)[rank]
        handle = None
        if mine_bg:
            level = [k for k, e in self._engines.items() if e is model._engine]
            tag = level[0] if level else "aux"
            aux = [self._engine("%s#%d" % (tag, j)) for j in range(1, min(conc, len(mine_bg)) + 1)]
            free = [p for p in model.parameters()]       # every parameter is free during the restarts
            handle = model.start_background_restarts(mine_bg, aux, free=free, rand_gen=self._restart_rng(),
                                                     max_iters=self.restart_max_iters)
        runs = []
        if rank == 0:
            model.optimize(max_iters=self.first_run_max_iters)
        model[".*Gaussian_noise"].unfix()
        model[".*Gaussian_noise"].constrain_positive()
        if rank == 0 and num_restarts > 0:
            r0 = model.optimize(max_iters=self.restart_max_iters)   # restart 0 continues from the current point
            if r0 is not None:
                runs.append((r0.f_opt, r0.x_opt, 0))
        if handle is not None:
            runs += handle.result()
        if size > 1:
            runs = [r for part in self.comm.allgather_object(runs) for r in part]
        if runs:
            best = min(runs, key=lambda r: (r[0], r[2]))
            model.optimizer_array = best[1]

    @staticmethod
    def assign_restarts(num_restarts, size):
        """-> per rank, the randomized restarts (indices 1..num_restarts-1) it runs beside its other work.
        Rank 0 starts with a load of two runs (first run + restart 0, sequential); every restart goes to the rank with
        the smallest load, the highest such rank on ties (rank 0's extra runs can only slow its chain down)."""
        load = [0] * size
        load[0] = 2
        out = [[] for _ in range(size)]
        for i in range(1, num_restarts):
            r = min(range(size), key=lambda k: (load[k], -k))
            out[r].append(i)
            load[r] += 1
        return out

    # ---- adaptation loop ---------------------------------------------------------------------------------
    def adapt_and_plot(self, plot_means: bool = False, plot_uncertainties: bool = False, plot_error: bool = False,
                       eps: float = 1e-8):
        """the entropy-reduction loop of src/abstractMFGP.py:317-359 without the drawing: per step
        maximise the predictive variance, predict on the 1000-point diagonal of the box, refit with the
        acquired point appended, stop early once |max variance| < eps."""
        X = np.linspace(self.lower_bound, self.upper_bound, 1000)
        self.mse_history, self.acquired_points, self.acquisition_values = [], [], []
        for i in range(self.adapt_steps):
            acquired_x, fopt = self.get_input_with_highest_uncertainty(self)
            means, uncertainties = self.predict(X)
            self.last_diagonal_prediction = (means, uncertainties)
            new_hf_X = np.vstack((self.hf_X, acquired_x))
            self.acquired_points.append(np.array(acquired_x))
            self.acquisition_values.append(fopt)
            if (plot_error or plot_uncertainties) and getattr(self, "X_test", None) is not None:
                self.mse_history.append(self.get_mse(self.X_test, self.Y_test))
            if getattr(self, "reoptimize", True):
                self.fit(new_hf_X)                       # the reference: full re-optimisation at N + 1 rows
            else:
                self.append_hf_point(acquired_x)         # hyper-parameters kept: O(N^2) rank-1 append on the device
            if np.abs(fopt) < self.eps:
                self.adapt_steps = i + 1
                print("Iteration stopped after {} iterations!".format(i + 1)
                      + " minimum uncertainty reached: {:e}".format(fopt))
                break

    def close(self):
        for e in self._engines.values():
            e.close()
        self._engines = {}
