"""Abstract multi-fidelity GP model: configuration, kernel construction, the data-driven low-fidelity
level, the hyper-parameter recipe and the entropy-reduction adaptation loop.

Keeps the PUBLIC surface of /root/reference/src/abstractMFGP.py -- constructor arguments (:12-33), the abstract API
(:35-49), `initialize_kernel` (:51-60), `get_NARGP_kernel` (:62-80), `initialize_lf_level` (:82-106), `adapt_lf`
(:108-122), `get_input_with_highest_uncertainty` (:124-129), `ARD` (:131-137) and `adapt_and_plot` (:275-359) -- so
that models written against the reference keep working; the bodies are this package's own, organised around three
things the reference does not have: device-resident engine handles per fidelity level, concurrent / rank-sharded
restarts, and a rank-1 append alternative to refitting.  Every GPy call is replaced by the HIP-backed objects of
engine.py.  Plotting (matplotlib, :139-273 and the drawing branches of :290-352) is out of scope; the loop records
the quantities it would have drawn (`mse_history`, `acquired_points`, `acquisition_values`).
"""
import abc

import numpy as np

from . import engine as gp
from .adaptation_maximizers import AbstractMaximizer
from .sharding import LocalComm


def _unit_box_where_missing(lower, upper, dim):
    """the data bounds default to the unit box (src/abstractMFGP.py:28-33); a single missing side is filled too"""
    lo = np.zeros(dim) if lower is None else lower
    hi = np.ones(dim) if upper is None else upper
    return lo, hi


class AbstractMFGP(metaclass=abc.ABCMeta):

    # the constants the reference hard-codes, exposed as overridable attributes (SURVEY.md section 5 "Config")
    noise_ratio = 0.01          # sigma_n^2 := 0.01 Var(Y) before the first run (src/abstractMFGP.py:132)
    first_run_max_iters = 500   # :134
    restart_max_iters = 1000    # :137
    num_restarts = 6            # src/MFDataFusion.py:100
    lf_max_iters = 1000         # lf_model.optimize() default budget (src/abstractMFGP.py:103)
    eval_cap = None             # hard cap on objective evaluations per L-BFGS-B run (benchmarks: exact budgets)
    restart_concurrency = 1     # 1: the reference's sequential order (the default: this layer then drives the engine call for call like the reference); >1: that many randomized restarts run concurrently with the first run / restart 0 -- same runs, same winner, 1.7-2 x faster fits at N = 256 .. 2048 (tools/midsize_fit.py)
    restart_lockstep = None     # the 1 + num_restarts runs of a fit as LOCK-STEPPED runs: every round one batched pass evaluates all
                                # live runs of a lane (engine.LockstepLane, mfgp_eval_batch) -- same runs, same steps bit for bit, same
                                # winner as the sequential order.  None = True: below N ~ 6144 one evaluation leaves most of the GPU
                                # idle and fits are 1.5-3 x faster; at N = 8192 every mode delivers the same ~10.5 ms per evaluation
                                # at the socket's power cap (lock step by run generators 1 762-1 773 ms per bench step, round 3's
                                # concurrent threads on two auxiliary handles 1 770-1 772 on the same box, alternating; with a thread
                                # per run lock step had been 1-1.5 % behind).  False: the reference's sequential order, or
                                # `restart_concurrency` > 1 concurrent restarts.  An engine without eval_batch (a test double) gets
                                # the reference's sequential order.
    lockstep_width = None       # live slots per rank.  None: from N = 6144 half the rank's runs, rounded up -- 4 for the recipe's 1 + 6 runs:
                                # the sequential pair first run -> restart 0 in one slot, the five randomized restarts 2 + 2 + 1 in
                                # three more, so every round carries 4, then 3 evaluations (a pass is convex in its size there: 6 then
                                # 1 costs more); below, where a pass is bound by the serial chain and nearly flat in its size, all runs
                                # at once (6 slots on two lanes: N = 512 55 -> 37 ms, 2048 140 -> 121, profiles/r04_midsize_fit.txt)
    lockstep_lanes = None       # engine handles the live slots are dealt to (each lane batches ITS slots' evaluations: a lane's serial
                                # chain then overlaps another lane's bulk work, and the hosts' L-BFGS-B steps of one lane the GPU pass
                                # of another).  None: 2 for 768 <= N < 6144 (fits 8-10 % faster than on one lane at N = 1024 .. 4096,
                                # profiles/r04_midsize_fit.txt), 1 below (a second lane's thread costs more than it overlaps) and
                                # from 6144 (N = 8192: the job sits at the socket's power cap either way; one lane of 4 evaluations
                                # measured 1 % ahead of 2 x 2)
    shard_sequential = True     # multi-GPU: the fit's SEQUENTIAL evaluations -- the low-fidelity run (all ranks would idle through it)
                                # and first run -> restart 0 (beside the ranks that were dealt no restart) -- are shared by a group
                                # of ranks (mfgp_eval_sharded: same numbers bit for bit, the rows of L^-T / K^-1 split); needs RCCL
    last_fit_info = None        # how the last fit's runs were driven: driver (lbfgsb-generators / thread-per-run / sequential), lanes, matrix
                                # sets per batched pass per lane (sized from the device's free memory), out-of-memory fallbacks taken
    lockstep_threads = False    # True: the lock-stepped runs as one THREAD each on scipy's blocking fmin_l_bfgs_b (the form that does not
                                # need scipy's private L-BFGS-B core; taken by itself when that core cannot be driven by reverse
                                # communication -- lbfgsb.available()); False: one loop per lane over run generators (same runs, same steps)
    restart_lend_main = False   # the main engine joins the restarts' pool once its sequential runs are through
    restart_aux = None          # auxiliary engine handles of the concurrent restarts (None: restart_concurrency of them)
    diagonal_points = 1000      # resolution of the box diagonal the adaptation loop predicts on every step (:318)

    @abc.abstractmethod
    def __init__(self, name: str, input_dim: int, num_derivatives: int, tau: float, f_exact: callable,
                 lower_bound: np.ndarray, upper_bound: np.ndarray, f_low: callable, lf_X: np.ndarray, lf_Y: np.ndarray,
                 lf_hf_adapt_ratio: int, use_composite_kernel: bool, adapt_maximizer: AbstractMaximizer, eps: float):
        super().__init__()
        self.name, self.input_dim = name, input_dim
        self.num_derivatives, self.tau = num_derivatives, tau
        self.f_exact, self.f_low = f_exact, f_low
        self.lf_hf_adapt_ratio, self.adapt_maximizer, self.eps = lf_hf_adapt_ratio, adapt_maximizer, eps
        self.lower_bound, self.upper_bound = _unit_box_where_missing(lower_bound, upper_bound, input_dim)
        # not in the reference: rank plumbing, seeded restarts, one engine handle per fidelity level
        self.comm = LocalComm()
        self.seed = None
        self._fit_count = 0
        self._engines = {}

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

    def _level_of(self, model):
        """name under which the model's engine handle is registered (auxiliary handles are derived from it)"""
        for level, eng in self._engines.items():
            if eng is model._engine:
                return level
        return "aux"

    # ---- kernels -------------------------------------------------------------------------------------
    def _augmented_columns(self):
        """-> (input columns, augmentation columns) of the high-fidelity level's design matrix [X | stencil values]"""
        d = self.input_dim
        c = self.augm_iterator.new_entries_count()
        return np.arange(d), np.arange(d, d + c)

    def initialize_kernel(self, use_composite_kernel: bool):
        """self.kernel := the composite NARGP kernel, or one isotropic RBF over all d + c augmented columns (the
        reference passes no ARD flag either, src/abstractMFGP.py:59-60).  Built once per model object: every refit
        reuses it, so hyper-parameters warm-start (src/MFDataFusion.py:69,96)."""
        if use_composite_kernel:
            self.kernel = self.get_NARGP_kernel()
            return
        cols_in, cols_aug = self._augmented_columns()
        self.kernel = gp.RBF(len(cols_in) + len(cols_aug))

    def get_NARGP_kernel(self, kern_class1=gp.RBF, kern_class2=gp.RBF, kern_class3=gp.RBF):
        """k1(augmentation columns) * k2(input columns) + k3(input columns): the correlation between fidelities
        modulated over the input space, plus a bias term in the inputs alone (src/abstractMFGP.py:62-80; the three
        class hooks select the stationary family of each factor)."""
        cols_in, cols_aug = self._augmented_columns()
        cross_fidelity = kern_class1(len(cols_aug), active_dims=cols_aug)
        modulation = kern_class2(len(cols_in), active_dims=cols_in)
        bias = kern_class3(len(cols_in), active_dims=cols_in)
        return cross_fidelity * modulation + bias

    # ---- low-fidelity level ----------------------------------------------------------------------------
    def initialize_lf_level(self, f_low: callable = None, lf_X: np.ndarray = None, lf_Y: np.ndarray = None):
        """The low-fidelity level is EITHER a python function `f_low` OR a GP trained here on (lf_X, lf_Y) whose
        posterior mean then plays the role of f_low (src/abstractMFGP.py:82-106) -- never both, never neither."""
        from_function = f_low is not None
        from_data = lf_X is not None and lf_Y is not None and self.lf_hf_adapt_ratio is not None
        assert from_function != from_data, \
            "the low-fidelity level needs exactly one source: a prediction function f_low, or training data lf_X / lf_Y"
        self.data_driven_lf_approach = from_data
        if from_function:
            self.f_low = f_low
            return
        self.lf_X, self.lf_Y = lf_X, lf_Y
        self.lf_model = self._new_lf_model()
        self._optimize_on_rank0(self.lf_model, lambda: self.lf_model.optimize(max_iters=self.lf_max_iters))
        self.f_low = self._lf_posterior_mean

    def _optimize_on_rank0(self, model, run):
        """a sequential optimisation every rank needs the result of: rank 0 runs it, the others adopt its optimum (one
        small object broadcast) and factorise once at it when they next need the level -- instead of N identical
        L-BFGS-B runs.  (It stays on every rank's critical path: nothing of the next level can start without it.)"""
        if self.comm.size == 1:
            run()
            return
        group = self._group(model, list(range(self.comm.size)))
        if group is not None:
            group.run(model, run)          # ONE optimiser (rank 0), every evaluation shared by all ranks
        elif self.comm.rank == 0:
            run()
        model.optimizer_array = self.comm.bcast_object(model.optimizer_array if self.comm.rank == 0 else None, src=0)

    def _group(self, model, members):
        """the ShardGroup of `members` on `model`'s engine, or None (no RCCL, a test double, shard_sequential off, a group of
        one); formed once per (engine, members) and kept -- the call is collective over all ranks"""
        if not self.shard_sequential or len(members) < 2 or not hasattr(self.comm, "shard_group"):
            return None
        if getattr(self.comm, "transport", None) != "rccl":
            return None                    # (the collectives of a shared evaluation are RCCL's; a TCP-only job keeps rank 0's own runs)
        # keyed by the engine OBJECT (kept in the entry and compared with `is`: an id() may be reused once a handle is closed and
        # replaced) -- the call is collective, so a group that could not be formed is remembered too (every rank asks again at the
        # same point or none does), but it is said once, and close() forgets everything
        cache = self.__dict__.setdefault("_shard_groups", [])
        for eng, mem, group in cache:
            if eng is model._engine and mem == tuple(members):
                return group
        group = self.comm.shard_group(model._engine, members)
        cache.append((model._engine, tuple(members), group))
        if group is None and self.comm.rank in members and not self.__dict__.get("_shard_group_warned"):   # (None is also what a rank OUTSIDE the group gets)
            self.__dict__["_shard_group_warned"] = True
            import warnings
            warnings.warn("multi-GPU: the ranks %s could not form an RCCL group on this level's engine (%s): its sequential evaluations "
                          "run on rank 0 alone" % (list(members), getattr(self.comm, "rccl_error", None) or "no reason recorded"),
                          RuntimeWarning, stacklevel=2)
        return group

    def _lf_posterior_mean(self, t):
        """f_low of a data-driven level: the CURRENT low-fidelity GP's posterior mean (mean only: the O(N^2 N*) variance
        product is never asked for here); looked up per call because adapt_lf replaces self.lf_model"""
        return self.lf_model.predict_mean(t)

    def _new_lf_model(self):
        model = gp.GPRegression(X=self.lf_X, Y=self.lf_Y, initialize=True, engine=self._engine("lf"))
        model.eval_cap = self.eval_cap
        return model

    def adapt_lf(self):
        """Grow the low-fidelity training set where the low-fidelity GP itself is most uncertain, refitting it after
        every acquisition; `adapt_steps * lf_hf_adapt_ratio` points.  (This is what the docstring of the reference's
        version describes; its code, src/abstractMFGP.py:108-122, is unreachable -- MFDataFusion.adapt calls a
        name-mangled attribute that does not exist.)"""
        assert hasattr(self, 'lf_model'), "adapt_lf needs a data-driven low-fidelity level"
        for _ in range(self.adapt_steps * self.lf_hf_adapt_ratio):
            where, _ = self.adapt_maximizer.maximize(self.lf_model.predict, self.lower_bound, self.upper_bound)
            value = self.lf_model.predict(where[None])[0][0]     # the level has no external truth: its own mean
            self.lf_X = np.vstack((self.lf_X, where))
            self.lf_Y = np.vstack((self.lf_Y, value))
            self.lf_model = self._new_lf_model()
            self.ARD(self.lf_model, self.num_restarts)

    def get_input_with_highest_uncertainty(self, model=None):
        """-> (x, f_opt): the maximiser of THIS model's predictive variance over the box and the (negated) value the
        maximiser reports (src/abstractMFGP.py:124-129; `model` is accepted and unused there as well)"""
        return self.adapt_maximizer.maximize(self.predict, self.lower_bound, self.upper_bound)

    # ---- hyper-parameter recipe ------------------------------------------------------------------------
    def _restart_rng(self):
        if self.seed is None:
            return None  # the reference draws from the global numpy RNG
        fit_id = self._fit_count
        return lambda i: np.random.default_rng([int(self.seed), fit_id, i]).normal

    def _pin_noise(self, model):
        """noise variance := noise_ratio * Var(Y), held fixed (src/abstractMFGP.py:132-133)"""
        model[".*Gaussian_noise"] = model.Y.var() * self.noise_ratio
        model[".*Gaussian_noise"].fix()
        model.eval_cap = self.eval_cap

    @staticmethod
    def _free_noise(model):
        """(src/abstractMFGP.py:135-136)"""
        model[".*Gaussian_noise"].unfix()
        model[".*Gaussian_noise"].constrain_positive()

    def ARD(self, model, num_restarts):
        """The reference's recipe (src/abstractMFGP.py:131-137): pin the noise at 1 % of Var(Y), one L-BFGS-B run
        (500), free the noise, `num_restarts` restarts (1000 each), the best run wins: 1 + num_restarts runs per fit.

        With `restart_concurrency` > 1 the randomized restarts 1.. (which, by paramz' semantics, start from fresh
        N(0,1) draws and so do not depend on the first run) execute on auxiliary engine handles in background
        threads WHILE the main thread does the first run and restart 0; same runs, same winner rule."""
        self._pin_noise(model)
        batched = num_restarts >= 2 and hasattr(model._engine, "eval_batch")
        lock = True if self.restart_lockstep is None else bool(self.restart_lockstep)
        if batched and lock:
            self._ard_lockstep(model, num_restarts)
            return
        self.last_fit_info = {"driver": "sequential (the reference's call order)" if int(self.restart_concurrency) <= 1
                              else "concurrent restarts on auxiliary handles", "lanes": 1, "sets_per_pass": [0], "oom_fallbacks": []}
        if int(self.restart_concurrency) <= 1:
            model.optimize(max_iters=self.first_run_max_iters)
            self._free_noise(model)
            model.optimize_restarts(num_restarts, optimizer="bfgs", max_iters=self.restart_max_iters, verbose=False,
                                    rand_gen=self._restart_rng(), comm=self.comm)
            return
        self._ard_concurrent(model, num_restarts, int(self.restart_concurrency))

    def _ard_lockstep(self, model, num_restarts):
        """The recipe's runs in lock step on the model's own engine handle.  Slot 0 (rank 0 only) is the sequential piece --
        first run (noise pinned), then restart 0, which continues from it -- the other slots are this rank's randomized
        restarts (as in _ard_concurrent: `assign_restarts`); every round is one batched pass over the live slots."""
        rank, size = self.comm.rank, self.comm.size
        assign = self.assign_restarts(num_restarts, size)
        mine_bg = assign[rank]
        # the ranks that were dealt no restart share the sequential pair's evaluations with rank 0 (8 GPUs, 6 restarts: ranks 1-2)
        chain_members = [0] + [r for r in range(1, size) if not assign[r]]
        chain = self._group(model, chain_members) if (size > 1 and not assign[0]) else None
        if chain is not None:
            runs = []
            if chain.leads:
                chain.run(model, lambda: model.optimize(max_iters=self.first_run_max_iters))
                self._free_noise(model)
                r0 = chain.run(model, lambda: model.optimize(max_iters=self.restart_max_iters))
                if r0 is not None:
                    runs.append((r0.f_opt, r0.x_opt, 0))
            else:
                chain.run(model, None)
                self._free_noise(model)
                chain.run(model, None)
            runs = [r for part in self.comm.allgather_object(runs) for r in part]
            # (the ranks outside the group ran their restarts through the code below: every rank reaches the same two gathers)
            self._install_winner(model, runs, rank, size)
            return
        own = 1 if rank == 0 else 0
        n_runs = len(mine_bg) + 2 * own
        # (a rank without the sequential pair holds runs of equal length: all of them live -- two rounds of B / 2 cost more than one of B)
        width = int(self.lockstep_width) if self.lockstep_width else (
            n_runs if not own else ((n_runs + 1) // 2 if len(model.X) >= 6144 else max(n_runs - 1, 1)))
        n_bg_slots = min(len(mine_bg), max(width - own, 1 if mine_bg else 0))
        n_slots = n_bg_slots + own
        # deal the slots to the lanes: lane 0 = the model's own handle (slot 0, the sequential pair, lives there)
        # (lanes by size, measured with the run generators: below N ~ 768 a pass is so short that a second lane only adds its thread's
        # hand-offs -- N = 256: 19 ms on one lane, 30-36 on two; 512: 25-27 / 24-31 -- from 1024 the second lane's passes fill the first
        # one's idle phases: 1024: 49 / 44, 2048: 141 / 130, 4096: 586 / 544; at N = 8192 one lane measured 1 % ahead)
        want_lanes = int(self.lockstep_lanes) if self.lockstep_lanes else (2 if 768 <= len(model.X) < 6144 else 1)
        n_lanes = max(1, min(want_lanes, n_slots))
        # memory policy (round 5): a further lane is a further handle with a slab of its own (one matrix set: 32 Np^2 bytes) -- only
        # where the device has room for it beside the batches
        per_set, free0 = self._device_memory(model._engine)
        if per_set and n_lanes > 1 and free0 - self.memory_reserve(model._engine) < (n_lanes - 1 + n_slots) * per_set:
            n_lanes = 1
        # (round-robin; giving lane 0 the sequential pair ALONE was measured twice and is within the run-to-run spread: its single
        # evaluations then share the GPU with the other lane's batch of five -- N = 2048: 126 / 121 ms per fit with a thread per run,
        # 124 / 130 with the run generators; 4096: 592 / 587, 527 / 544; 1024: 47 / 44.  Stream priorities do not rescue it: with the
        # restarts' handle one level down (chain normal / bulk low beside chain high / bulk normal) the sequential pair's evaluations
        # speed up (N = 2048: 1.31 -> 1.1 ms) and the background lane starves: 124 -> 193 ms per fit, 4096: 539 -> 628.  Nor does a CU
        # mask on the restarts' handle (6, 5 or 4 of every 8 CUs): the sequential pair's lane gains what the masked lane loses and the
        # masked lane becomes the longer one -- N = 2048: 115 -> 120 / 131 / 152 ms, 4096: 533 -> 587 / 661 / 780, 1024: 54 -> 45-48)
        per_lane = [[k for k in range(n_slots) if k % n_lanes == j] for j in range(n_lanes)]
        tag = self._level_of(model)
        engines = [model._engine] + [self._engine("%s#%d" % (tag, j)) for j in range(1, n_lanes)]
        for e in engines[1:]:
            e.set_data(model.X, model.Y[:, 0])
            e.set_kernel(model._parts)
        # ... and a lane's passes carry as many sets as fit (the lanes share what is free; a lane that gets fewer sets than it has
        # slots runs its rounds in chunks, one that gets none evaluates request by request on its handle's own slab -- the same
        # steps, the same fit, bit for bit: LockstepLane)
        budgets = self._lane_budgets(engines, [len(sl) for sl in per_lane])
        self.last_fit_info = {"driver": "lbfgsb-generators", "lanes": n_lanes, "slots_per_lane": [len(sl) for sl in per_lane],
                              "sets_per_pass": budgets, "oom_fallbacks": []}
        if gp._lbfgsb.available() and not self.lockstep_threads:
            self._lockstep_by_programs(model, engines, per_lane, mine_bg, own, rank, size, budgets)
            return
        self.last_fit_info["driver"] = "thread-per-run (scipy's L-BFGS-B core not drivable by reverse communication)" \
            if not self.lockstep_threads else "thread-per-run (lockstep_threads)"
        lanes = [gp.LockstepEvaluator(e, len(sl)) for e, sl in zip(engines, per_lane)]
        lockstep = lanes[0] if lanes else None
        handle = None
        if mine_bg:
            bg = [(ls, [k for k in sl if not (own and k == 0)]) for ls, sl in zip(lanes, per_lane)]
            handle = model.start_lockstep_restarts(mine_bg, bg, free=list(model.parameters()),      # all free during the restarts
                                                   rand_gen=self._restart_rng(), max_iters=self.restart_max_iters)
        runs = []
        try:
            if rank == 0:
                model._eval_hook = lambda th, nz, jit: lockstep.evaluate(0, th, nz, jit)
                model.optimize(max_iters=self.first_run_max_iters)
            self._free_noise(model)
            if rank == 0:
                r0 = model.optimize(max_iters=self.restart_max_iters)   # restart 0 continues from the current point
                if r0 is not None:
                    runs.append((r0.f_opt, r0.x_opt, 0))
        finally:
            if rank == 0:
                lockstep.retire(0)
                model._eval_hook = None
                model._dirty = True        # the handle's own factorisation was never at the points the batches evaluated
                model._have_grad = False
        if handle is not None:
            runs += handle.result()
        self.last_lockstep = lockstep
        self.last_lockstep_lanes = lanes
        if size > 1:
            runs = [r for part in self.comm.allgather_object(runs) for r in part]
        self._install_winner(model, runs, rank, size)

    # ---- memory policy of the batched evaluations --------------------------------------------------------
    @staticmethod
    def _device_memory(engine):
        """(bytes of one matrix set of a batch on `engine`, free bytes of its device) -- (0, 0) for an engine that cannot say (test doubles)"""
        if not (hasattr(engine, "batch_mem") and hasattr(engine, "mem_info")):
            return 0, 0
        return int(engine.batch_mem(1)[0]), int(engine.mem_info()[0])

    @staticmethod
    def memory_reserve(engine):
        """device memory a fit leaves alone: 2 % of the device, at least 1 GiB (plan tables, predictive panels, the host application)"""
        total = int(engine.mem_info()[1]) if hasattr(engine, "mem_info") else 0
        return max(1 << 30, total // 50)

    def _lane_budgets(self, engines, slots):
        """matrix sets each lane's passes may carry: its slot count where everything fits; otherwise the free device memory (less the
        reserve) dealt to the lanes in proportion to their slots, plus what a lane's handle holds already, and within
        MFGP_BATCH_MEM_CAP; 0 = request by request.  None per lane for an engine that cannot say."""
        if not engines or not hasattr(engines[0], "batch_mem"):
            return [None] * len(engines)
        info = [e.batch_mem(1) for e in engines]                     # (bytes per set, cap bytes, sets held)
        free = int(engines[0].mem_info()[0]) - self.memory_reserve(engines[0])
        need = sum(max(0, n - held) * per for n, (per, _, held) in zip(slots, info))
        out = []
        for n, (per, cap, held) in zip(slots, info):
            fit = n if need <= free else held + int(max(0, free) * (n / max(1, sum(slots))) // per)
            if cap:
                fit = min(fit, cap // per)
            out.append(int(max(0, min(n, fit))))
        return out

    def _lockstep_by_programs(self, model, engines, per_lane, mine_bg, own, rank, size, budgets=None):
        """The lock-stepped runs WITHOUT a thread per run: every run is a generator of engine evaluation requests
        (engine.GPRegression._run_gen: scipy's L-BFGS-B core by reverse communication), every lane one loop over its runs
        (engine.LockstepLane.drive) -- lane 0 in the calling thread, each further lane in one thread of its own, so that one lane's
        L-BFGS-B steps run beside the other's batched pass.  Slot 0's program is the sequential pair: first run (noise pinned), the
        noise freed, restart 0 from there."""
        import collections
        import threading
        free_all = list(model.parameters())                 # every parameter is free during the restarts
        starts = model.restart_starts(mine_bg, free_all, self._restart_rng())
        todo = collections.deque(mine_bg)
        todo_lock = threading.Lock()

        def take_index():
            with todo_lock:
                return todo.popleft() if todo else None

        out, first = {}, []

        def sequential_pair():
            yield from model.optimize_program(self.first_run_max_iters)
            self._free_noise(model)
            yield from model.optimize_program(self.restart_max_iters, first)     # restart 0 continues from the current point

        if not own:
            self._free_noise(model)
        budgets = budgets or [None] * len(engines)
        lanes = [gp.LockstepLane(e, max_batch=(None if b is None else max(1, b))) for e, b in zip(engines, budgets)]
        programs = [[sequential_pair() if (own and k == 0) else model.restart_program(take_index, starts, free_all, self.restart_max_iters, out)
                     for k in slots] for slots in per_lane]
        errors = []

        def drive(lane, progs):
            try:
                lane.drive(progs)
            except BaseException as ex:  # noqa: BLE001 - re-raised by the calling thread below
                errors.append(ex)

        threads = [threading.Thread(target=drive, args=(ln, pr), daemon=True) for ln, pr in zip(lanes[1:], programs[1:])]
        for t in threads:
            t.start()
        try:
            lanes[0].drive(programs[0])
        finally:
            for t in threads:
                t.join()
            model._dirty = True        # no handle's own factorisation is at a point the batches evaluated
            model._have_grad = False
        if errors:
            raise errors[0]
        runs = [(first[0].f_opt, first[0].x_opt, 0)] if first else []
        runs += [out[i] for i in mine_bg]
        self.last_fit_info["sets_per_pass_used"] = [ln.max_batch for ln in lanes]
        self.last_fit_info["oom_fallbacks"] = [list(ln.oom_fallbacks) for ln in lanes]
        self.last_lockstep = lanes[0]
        self.last_lockstep_lanes = lanes
        if size > 1:
            runs = [r for part in self.comm.allgather_object(runs) for r in part]
        self._install_winner(model, runs, rank, size)

    def _install_winner(self, model, runs, rank, size):
        if runs:
            best = min(runs, key=lambda r: (r[0], r[2]))
            model.optimizer_array = best[1]
        elif size > 1:
            model.optimizer_array = self.comm.bcast_object(model.optimizer_array if rank == 0 else None, src=0)

    def _ard_concurrent(self, model, num_restarts, conc):
        # Rank 0 owns the only sequential piece (first run -> restart 0, which continues from it); the randomized
        # restarts go to the least-loaded rank.  No other rank needs the first run: the winner overwrites every
        # free parameter on every rank.
        rank, size = self.comm.rank, self.comm.size
        mine_bg = self.assign_restarts(num_restarts, size)[rank]
        handle = None
        if mine_bg:
            tag = self._level_of(model)
            n_aux = self.restart_aux if self.restart_aux is not None else conc
            aux = [self._engine("%s#%d" % (tag, j)) for j in range(1, min(n_aux, len(mine_bg)) + 1)]
            handle = model.start_background_restarts(mine_bg, aux, free=list(model.parameters()),   # all free during restarts
                                                     rand_gen=self._restart_rng(), max_iters=self.restart_max_iters,
                                                     spare=1 if self.restart_lend_main else 0)
        runs = []
        if rank == 0:
            model.optimize(max_iters=self.first_run_max_iters)
        self._free_noise(model)
        if rank == 0 and num_restarts > 0:
            r0 = model.optimize(max_iters=self.restart_max_iters)   # restart 0 continues from the current point
            if r0 is not None:
                runs.append((r0.f_opt, r0.x_opt, 0))
        if handle is not None:
            if self.restart_lend_main:
                # this rank's sequential share is through (rank 0: first run -> restart 0; the others had none): its main engine
                # joins the pool, so that the last restarts do not run alone on one auxiliary handle
                model.lend_engine(handle)
            runs += handle.result()
        if size > 1:
            runs = [r for part in self.comm.allgather_object(runs) for r in part]
        if runs:
            best = min(runs, key=lambda r: (r[0], r[2]))
            model.optimizer_array = best[1]
        elif size > 1:
            # no restart produced a result (num_restarts = 0, or every run ended without one): rank 0 keeps the optimum of its
            # first run and the other ranks -- which never ran it -- adopt that, or the replicated level state would diverge
            model.optimizer_array = self.comm.bcast_object(model.optimizer_array if rank == 0 else None, src=0)

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
        """The entropy-reduction loop of src/abstractMFGP.py:317-359 without the drawing.  One step = maximise the
        predictive variance over the box, predict on the box diagonal (what the reference plots), take the maximiser
        as a new high-fidelity point, and either refit from scratch (the reference) or append at fixed
        hyper-parameters (`reoptimize=False`).  Stops early once the reported maximum falls below `self.eps`."""
        diagonal = np.linspace(self.lower_bound, self.upper_bound, self.diagonal_points)
        track_error = (plot_error or plot_uncertainties) and getattr(self, "X_test", None) is not None
        self.mse_history, self.acquired_points, self.acquisition_values = [], [], []
        planned = self.adapt_steps
        for step in range(1, planned + 1):
            where, reported = self.get_input_with_highest_uncertainty(self)
            self.last_diagonal_prediction = self.predict(diagonal)
            self.acquired_points.append(np.array(where))
            self.acquisition_values.append(reported)
            if track_error:
                self.mse_history.append(self.get_mse(self.X_test, self.Y_test))
            if getattr(self, "reoptimize", True):
                self.fit(np.vstack((self.hf_X, where)))   # full re-optimisation at N + 1 rows
            else:
                self.append_hf_point(where)               # hyper-parameters kept: O(N^2) rank-1 append on the device
            if abs(reported) < self.eps:
                self.adapt_steps = step                   # callers read the number of acquisitions made (gpc driver)
                print("adaptation stopped after %d of %d steps: largest predictive variance %.3e is below eps = %.1e"
                      % (step, planned, abs(reported), self.eps))
                break

    # ---- plotting: out of scope -------------------------------------------------------------------
    def _no_plot(self, *args, **kwargs):
        """the reference draws with matplotlib (src/abstractMFGP.py:139-273, called from src/MethodAssessment.py:51-56); this
        package records what would have been drawn instead (`mse_history`, `acquired_points`, `acquisition_values`,
        `last_diagonal_prediction`) -- callers that ask for a figure are told so, not handed an AttributeError"""
        raise NotImplementedError("plotting is not part of this package (the reference's matplotlib code is out of scope): read "
                                  "model.mse_history / acquired_points / acquisition_values / last_diagonal_prediction and plot them "
                                  "with matplotlib in the calling code")

    plot = plot_forecast = plot_uncertainties_2D = plot_compare_with_exact = _no_plot      # the reference's public plot methods

    def close(self):
        # the communicators of the groups this object formed (collective: every member of a group closes); the world communicator
        # attached by comm.attach_engine is the caller's, as the engines passed in from outside are
        for eng, _, group in self.__dict__.pop("_shard_groups", []):
            if group is not None and getattr(eng, "comm_size", 1) > 1 and eng is not getattr(self.comm, "_engine", None) \
                    and not getattr(eng, "comm_aborted", False):
                try:
                    eng.comm_destroy()
                except Exception:  # noqa: BLE001 - closing: a handle that is already gone
                    pass
        for e in self._engines.values():
            e.close()
        self._engines = {}
