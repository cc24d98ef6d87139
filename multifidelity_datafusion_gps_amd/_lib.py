"""ctypes binding of libmfgp_hip.so (the C-ABI declared in include/mfgp.h).

This is the whole Python<->HIP boundary: plain pointers and sizes, no torch types.  The library is
the product -- there is no CPU fallback: if it is missing or no HIP device is present, loading /
`Engine()` raises `EngineUnavailable` loudly.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmfgp_hip.so")

KERN_RBF, KERN_MATERN32, KERN_MATERN52 = 0, 1, 2
KERN_ARD = 0x100      # OR-ed into a part's type: one lengthscale per active column (include/mfgp.h)
MAX_PARTS = 6
MAX_THETA = 40


def num_params(parts):
    """P of a kernel description [(type, col_begin, col_end, term)]: per factor a variance + 1 (or, ARD, one per column) lengthscale"""
    return sum(1 + ((int(p[2]) - int(p[1])) if (int(p[0]) & KERN_ARD) else 1) for p in parts)

# every symbol include/mfgp.h declares (tests check the .so exports each of them)
EXPORTED_SYMBOLS = [
    "mfgp_create", "mfgp_destroy", "mfgp_last_error", "mfgp_device_info", "mfgp_build_id", "mfgp_set_data",
    "mfgp_set_kernel", "mfgp_num_params", "mfgp_eval", "mfgp_eval_batch", "mfgp_mem_info", "mfgp_batch_mem", "mfgp_eval_sharded", "mfgp_sharded_lead", "mfgp_sharded_serve", "mfgp_sharded_release", "mfgp_kbuild_rows", "mfgp_kbuild_owned_rows", "mfgp_row_block_owner", "mfgp_dev_matrix", "mfgp_eval_prebuilt", "mfgp_factorize", "mfgp_nlml", "mfgp_nlml_grad", "mfgp_append_row", "mfgp_predict",
    "mfgp_augment", "mfgp_predict_chained",
    "mfgp_get_K", "mfgp_get_L", "mfgp_get_Linv", "mfgp_get_Kinv", "mfgp_get_alpha", "mfgp_get_timings",
    "mfgp_get_counters", "mfgp_device_synchronize",
    "mfgp_comm_unique_id", "mfgp_comm_init", "mfgp_comm_destroy", "mfgp_comm_state", "mfgp_comm_calibrate", "mfgp_shard_decision", "mfgp_dist_cholesky_pays", "mfgp_allgather_rows", "mfgp_allgather_host",
    "mfgp_rows_download", "mfgp_rows_upload",
    "mfgp_dbg_gemm_nt", "mfgp_dbg_leaf", "mfgp_dbg_eval_as_rank", "mfgp_dbg_fail_sharded_after",
    "mfgp_dbg_fail_collective_after",
]


ERR_OOM = -6      # MFGP_ERR_OOM (include/mfgp.h)


class EngineOutOfMemory(RuntimeError):
    """mfgp_eval_batch: the batch's matrix sets do not fit the device (or MFGP_BATCH_MEM_CAP).  The handle is unchanged and usable:
    retry with fewer sets, or with eval(), which needs none (engine.LockstepLane does)."""


class EngineUnavailable(RuntimeError):
    """libmfgp_hip.so is missing / not loadable, or there is no HIP device."""


class NotPositiveDefinite(np.linalg.LinAlgError):
    """The Cholesky met a non-positive pivot (status > 0 = 1-based pivot index)."""

    def __init__(self, info, msg=""):
        super().__init__(msg or "not positive definite (pivot %d)" % info)
        self.info = info


class KernPart(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int32), ("col_begin", ctypes.c_int32), ("col_end", ctypes.c_int32),
                ("term", ctypes.c_int32)]


class Timings(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in (
        "kbuild_ms", "cholinv_ms", "solve_ms", "kinv_ms", "grad_ms", "predict_panel_ms", "predict_var_ms",
        "total_ms", "kbuild_bytes", "kinv_flops", "cholinv_flops")] + [("n_launches", ctypes.c_int64), ("timed", ctypes.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Counters(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in (
        "evals", "grad_evals", "predicts", "predict_rows", "kbuild_ms", "cholinv_ms", "solve_ms", "kinv_ms",
        "grad_ms", "total_ms", "predict_ms", "kbuild_bytes", "kinv_flops", "cholinv_flops", "predict_panel_ms",
        "predict_var_ms", "predict_var_flops", "timed_evals", "timed_predict_var_flops")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def load_library(path=None):
    """dlopen the engine (once) and declare the prototypes."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise EngineUnavailable(
            "%s not found: build it with `python -m multifidelity_datafusion_gps_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % p)
    # HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues per priority (ROCm's default: 4).  For a fit that keeps
    # several engine handles busy, 2 per priority measured best (profiles/r03_hw_queues.txt: cfg3's HF level 404 -> 345 ms, the
    # N = 8192 bench 1777 -> 1765 ms).  OPT-IN: the setting is process-wide, changes the queue mapping of every other
    # HIP user in the host application, and only takes effect if it is in the environment before the HIP runtime initialises --
    # so it is applied only where the caller asks for it with MFGP_HW_QUEUES=<n> (bench.py and the tools do); a
    # GPU_MAX_HW_QUEUES already in the environment wins.  `hw_queues_setting()` reports what is in effect.
    if os.environ.get("MFGP_HW_QUEUES", "0") not in ("", "0"):
        os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ["MFGP_HW_QUEUES"])
    try:
        lib = ctypes.CDLL(p)
    except OSError as e:  # missing ROCm runtime etc.
        raise EngineUnavailable("cannot load %s: %s" % (p, e)) from e
    H = ctypes.c_void_p
    dp = ctypes.POINTER(ctypes.c_double)
    i32, i64, f64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_double
    protos = {
        "mfgp_create": (i32, [i32, ctypes.POINTER(H)]),
        "mfgp_destroy": (i32, [H]),
        "mfgp_last_error": (ctypes.c_char_p, [H]),
        "mfgp_device_info": (ctypes.c_char_p, [H]),
        "mfgp_build_id": (ctypes.c_char_p, []),
        "mfgp_set_data": (i32, [H, dp, i64, i32, dp]),
        "mfgp_set_kernel": (i32, [H, ctypes.POINTER(KernPart), i32]),
        "mfgp_num_params": (i32, [ctypes.POINTER(KernPart), i32]),
        "mfgp_eval": (i32, [H, dp, f64, f64, i32, dp, dp]),
        "mfgp_eval_batch": (i32, [H, i32, dp, dp, dp, i32, dp, dp, ctypes.POINTER(i32)]),
        "mfgp_mem_info": (i32, [H, ctypes.POINTER(i64), ctypes.POINTER(i64)]),
        "mfgp_batch_mem": (i32, [H, i32, ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(i32)]),
        "mfgp_eval_sharded": (i32, [H, dp, f64, f64, i32, dp, dp]),
        "mfgp_dbg_eval_as_rank": (i32, [H, dp, f64, f64, i32, i32, i32, dp]),
        "mfgp_sharded_lead": (i32, [H, dp, f64, f64, i32, dp, dp]),
        "mfgp_sharded_serve": (i32, [H, ctypes.POINTER(i64)]),
        "mfgp_sharded_release": (i32, [H]),
        "mfgp_kbuild_rows": (i32, [H, dp, f64, f64, i64, i64]),
        "mfgp_kbuild_owned_rows": (i32, [H, dp, f64, f64, i32, i32]),
        "mfgp_row_block_owner": (i32, [i32, i32]),
        "mfgp_dev_matrix": (i32, [H, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(i64)]),
        "mfgp_eval_prebuilt": (i32, [H, i32, dp, dp]),
        "mfgp_factorize": (i32, [H, dp, f64, f64]),
        "mfgp_nlml": (i32, [H, dp]),
        "mfgp_nlml_grad": (i32, [H, dp]),
        "mfgp_append_row": (i32, [H, dp, f64]),
        "mfgp_predict": (i32, [H, dp, i64, dp, dp, i32, i32]),
        "mfgp_augment": (i32, [H, dp, i64, dp, i32, dp]),
        "mfgp_predict_chained": (i32, [H, H, dp, i64, dp, i32, dp, dp, i32, i32, dp]),
        "mfgp_get_K": (i32, [H, dp]),
        "mfgp_get_L": (i32, [H, dp]),
        "mfgp_get_Linv": (i32, [H, dp]),
        "mfgp_get_Kinv": (i32, [H, dp]),
        "mfgp_get_alpha": (i32, [H, dp]),
        "mfgp_get_timings": (i32, [H, ctypes.POINTER(Timings)]),
        "mfgp_get_counters": (i32, [H, ctypes.POINTER(Counters), i32]),
        "mfgp_device_synchronize": (i32, [H]),
        "mfgp_comm_unique_id": (i32, [ctypes.POINTER(ctypes.c_uint8)]),
        "mfgp_comm_init": (i32, [H, ctypes.POINTER(ctypes.c_uint8), i32, i32]),
        "mfgp_comm_destroy": (i32, [H]),
        "mfgp_comm_state": (i32, [H]),
        "mfgp_comm_calibrate": (i32, [H, i32, i32, dp]),
        "mfgp_shard_decision": (i32, [H, dp]),
        "mfgp_dist_cholesky_pays": (i32, [i32, i32, ctypes.c_double, dp, dp]),
        "mfgp_dbg_fail_sharded_after": (i32, [H, i32]),
        "mfgp_dbg_fail_collective_after": (i32, [H, i32]),
        "mfgp_allgather_rows": (i32, [H]),
        "mfgp_allgather_host": (i32, [H, dp, i64, dp]),
        "mfgp_rows_download": (i32, [H, i64, i64, dp]),
        "mfgp_rows_upload": (i32, [H, i64, i64, dp]),
        "mfgp_dbg_gemm_nt": (i32, [H, dp, dp, dp, i32, i32, i32, f64, f64, i32]),
        "mfgp_dbg_leaf": (i32, [H, dp, dp, dp, dp]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def hw_queues_setting():
    """GPU_MAX_HW_QUEUES as this process will hand it to the HIP runtime ("runtime default" when unset)"""
    return os.environ.get("GPU_MAX_HW_QUEUES", "runtime default")


def build_id():
    """the source hash the loaded library was built from (mfgp_build_id)"""
    return load_library().mfgp_build_id().decode()


def _dptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Engine:
    """One engine handle = one HIP device + stream + the device-resident state of one GP level."""

    def __init__(self, device=None):
        self._lib = load_library()
        implicit = device is None
        if implicit:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        self._poisoned = None
        h = ctypes.c_void_p()
        rc = self._lib.mfgp_create(int(device), ctypes.byref(h))
        if rc != 0 and implicit and device != 0 and b"bad device id" in self._lib.mfgp_last_error(None):
            # one process per GPU under a launcher that narrows the visible devices per rank: the only device is 0
            device = 0
            rc = self._lib.mfgp_create(0, ctypes.byref(h))
        if rc != 0:
            msg = self._lib.mfgp_last_error(None).decode()
            self._handle = None
            raise EngineUnavailable("mfgp_create(device=%d) failed (%d): %s" % (device, rc, msg))
        self._handle = h
        self.device = device
        self.n = 0
        self.n_parts = 0
        self.n_params = 0      # P: kernel parameters of the description last given to set_kernel (theta length)
        self.comm_rank, self.comm_size = 0, 1

    # -- plumbing -------------------------------------------------------------------------------
    @property
    def _h(self):
        """the C handle; unusable once poisoned (see poison())"""
        if self._poisoned is not None:
            raise RuntimeError("engine handle abandoned: " + self._poisoned)
        return self._handle

    def poison(self, why):
        """give the handle up without destroying it: another thread is stuck inside a library call on it (an RCCL
        initialisation that never returned) and the handle is not thread-safe, so no further call may reach it --
        every method raises from here on, close() leaves it alone, and the process is expected to end."""
        self._poisoned = str(why)

    def close(self):
        if getattr(self, "_poisoned", None) is not None:
            return
        if getattr(self, "_handle", None):
            self._lib.mfgp_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, who):
        if rc == 0:
            return
        msg = self._lib.mfgp_last_error(self._h).decode()
        if rc > 0:
            raise NotPositiveDefinite(rc, "%s: %s" % (who, msg))
        if rc == ERR_OOM:
            raise EngineOutOfMemory("%s: %s" % (who, msg))
        raise RuntimeError("%s failed (%d): %s" % (who, rc, msg))

    @property
    def device_info(self):
        return self._lib.mfgp_device_info(self._h).decode()

    # -- state ----------------------------------------------------------------------------------
    def set_data(self, X, Y):
        X = _c64(X)
        Y = _c64(Y).reshape(-1)
        if X.ndim != 2 or Y.shape[0] != X.shape[0]:
            raise ValueError("X must be (N, D) and Y (N,) / (N, 1)")
        self._check(self._lib.mfgp_set_data(self._h, _dptr(X), X.shape[0], X.shape[1], _dptr(Y)), "mfgp_set_data")
        self.n, self.d = X.shape

    def set_kernel(self, parts):
        """parts: iterable of (type, col_begin, col_end, term)."""
        parts = list(parts)
        arr = (KernPart * len(parts))(*[KernPart(*map(int, p)) for p in parts])
        self._check(self._lib.mfgp_set_kernel(self._h, arr, len(parts)), "mfgp_set_kernel")
        self.n_parts = len(parts)
        self.n_params = int(self._lib.mfgp_num_params(arr, len(parts)))

    # -- hot calls ------------------------------------------------------------------------------
    def eval(self, theta, noise, jitter=1e-8, want_grad=True):
        theta = self._theta(theta)
        nlml = ctypes.c_double()
        grad = np.zeros(self.n_params + 1)
        rc = self._lib.mfgp_eval(self._h, _dptr(theta), float(noise), float(jitter), int(bool(want_grad)),
                                 ctypes.byref(nlml), _dptr(grad))
        self._check(rc, "mfgp_eval")
        return (nlml.value, grad) if want_grad else nlml.value

    def eval_sharded(self, theta, noise, jitter=1e-8, want_grad=True):
        """eval() as ONE evaluation across the ranks of this handle's communicator (collective; comm_init first; without a
        communicator: the group of one).  Same results as eval(), bit for bit, on every rank."""
        theta = self._theta(theta)
        nlml = ctypes.c_double()
        grad = np.zeros(self.n_params + 1)
        rc = self._lib.mfgp_eval_sharded(self._h, _dptr(theta), float(noise), float(jitter), int(bool(want_grad)),
                                         ctypes.byref(nlml), _dptr(grad))
        self._check(rc, "mfgp_eval_sharded")
        return (nlml.value, grad) if want_grad else nlml.value

    def sharded_lead(self, theta, noise, jitter=1e-8, want_grad=True):
        """rank 0 of the communicator: eval_sharded whose arguments travel to the serving ranks (sharded_serve) with the call"""
        theta = self._theta(theta)
        nlml = ctypes.c_double()
        grad = np.zeros(self.n_params + 1)
        rc = self._lib.mfgp_sharded_lead(self._h, _dptr(theta), float(noise), float(jitter), int(bool(want_grad)),
                                         ctypes.byref(nlml), _dptr(grad))
        self._check(rc, "mfgp_sharded_lead")
        return (nlml.value, grad) if want_grad else nlml.value

    def sharded_serve(self):
        """ranks > 0: run this rank's share of every evaluation the leader asks for; returns their number at the leader's release"""
        n = ctypes.c_int64()
        self._check(self._lib.mfgp_sharded_serve(self._h, ctypes.byref(n)), "mfgp_sharded_serve")
        return n.value

    def sharded_release(self):
        self._check(self._lib.mfgp_sharded_release(self._h), "mfgp_sharded_release")

    def dbg_eval_as_rank(self, theta, noise, rank, size, jitter=1e-8, want_grad=True):
        """-> milliseconds of the device work of rank `rank` of `size` of a sharded evaluation, without its exchange steps"""
        theta = self._theta(theta)
        ms = ctypes.c_double()
        self._check(self._lib.mfgp_dbg_eval_as_rank(self._h, _dptr(theta), float(noise), float(jitter), int(bool(want_grad)),
                                                    int(rank), int(size), ctypes.byref(ms)), "mfgp_dbg_eval_as_rank")
        return ms.value

    MAX_BATCH = 16

    def eval_batch(self, thetas, noises, jitters=1e-8, want_grad=True):
        """B evaluations at once (mfgp_eval_batch): thetas (B, P), noises (B,), jitters (B,) or one value ->
        (nlml (B,), grads (B, P + 1) or None, status (B,) int: 0 or the failed pivot's index).  Bitwise the results of B eval()
        calls; the handle's own factorisation is left alone."""
        thetas = _c64(thetas)
        if thetas.ndim != 2 or thetas.shape[1] != self.n_params:
            raise ValueError("thetas must be (B, %d)" % self.n_params)
        B = thetas.shape[0]
        noises = _c64(np.broadcast_to(np.asarray(noises, dtype=np.float64), (B,)))
        jitters = _c64(np.broadcast_to(np.asarray(jitters, dtype=np.float64), (B,)))
        nlml = np.empty(B)
        grads = np.zeros((B, self.n_params + 1)) if want_grad else None
        status = np.zeros(B, dtype=np.int32)
        rc = self._lib.mfgp_eval_batch(self._h, B, _dptr(thetas), _dptr(noises), _dptr(jitters), int(bool(want_grad)),
                                       _dptr(nlml), _dptr(grads) if want_grad else None,
                                       status.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        self._check(rc, "mfgp_eval_batch")
        return nlml, grads, status

    def mem_info(self):
        """(free, total) bytes of the handle's device"""
        f, t = ctypes.c_int64(), ctypes.c_int64()
        self._check(self._lib.mfgp_mem_info(self._h, ctypes.byref(f), ctypes.byref(t)), "mfgp_mem_info")
        return f.value, t.value

    def batch_mem(self, sets):
        """(bytes `sets` matrix sets of a batch take on this handle, MFGP_BATCH_MEM_CAP or 0, sets the handle holds already)"""
        b, c, held = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32()
        self._check(self._lib.mfgp_batch_mem(self._h, int(sets), ctypes.byref(b), ctypes.byref(c), ctypes.byref(held)), "mfgp_batch_mem")
        return b.value, c.value, held.value

    def batch_sets_that_fit(self, reserve_bytes=0):
        """how many matrix sets a batch on this handle may hold: by the free device memory (less `reserve_bytes`, plus what the handle's
        batch slab holds already) and by MFGP_BATCH_MEM_CAP; at most MAX_BATCH"""
        per, cap, held = self.batch_mem(1)
        free, _ = self.mem_info()
        fit = max(0, free - int(reserve_bytes)) // per + held
        if cap:
            fit = min(fit, cap // per)
        return int(min(fit, self.MAX_BATCH))

    # -- row-block K build + all-gather (multi-GPU layout of SURVEY 8(e3)) ---------------------------
    def _theta(self, theta):
        theta = _c64(theta).reshape(-1)
        if theta.shape[0] != self.n_params:     # the library reads exactly P entries
            raise ValueError("theta must have %d entries (variance + lengthscale(s) per part)" % self.n_params)
        return theta

    def kbuild_rows(self, theta, noise, jitter, row_begin, row_end):
        theta = self._theta(theta)
        self._check(self._lib.mfgp_kbuild_rows(self._h, _dptr(theta), float(noise), float(jitter), int(row_begin),
                                               int(row_end)), "mfgp_kbuild_rows")

    def kbuild_owned_rows(self, theta, noise, jitter, rank, size):
        """the rows of every 128-row block rank `rank` of `size` owns (row_block_owner): what a rank builds before allgather_rows"""
        theta = self._theta(theta)
        self._check(self._lib.mfgp_kbuild_owned_rows(self._h, _dptr(theta), float(noise), float(jitter), int(rank), int(size)),
                    "mfgp_kbuild_owned_rows")

    @staticmethod
    def row_block_owner(block, size):
        """owner of 128-row block `block` among `size` ranks: serpentine block-cyclic (0 1 .. G-1 G-1 .. 1 0 ..)"""
        return int(load_library().mfgp_row_block_owner(int(block), int(size)))

    def dev_matrix(self):
        """(device pointer, padded size Np) of the Np x Np fp64 matrix the factorisation consumes"""
        p, n = ctypes.c_void_p(), ctypes.c_int64()
        self._check(self._lib.mfgp_dev_matrix(self._h, ctypes.byref(p), ctypes.byref(n)), "mfgp_dev_matrix")
        return p.value, n.value

    # -- exchange steps of the multi-GPU path (RCCL inside the library; comm_rccl.hip) ------------------
    @staticmethod
    def comm_unique_id():
        """128 opaque bytes (ncclUniqueId) to be produced on rank 0 and handed to every rank's comm_init"""
        lib = load_library()
        buf = (ctypes.c_uint8 * 128)()
        rc = lib.mfgp_comm_unique_id(buf)
        if rc != 0:
            raise RuntimeError("mfgp_comm_unique_id failed (%d): %s" % (rc, lib.mfgp_last_error(None).decode()))
        return bytes(buf)

    def comm_init(self, unique_id, rank, size):
        """collective: bind an RCCL communicator of `size` ranks to this handle (its device, its stream)"""
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self._lib.mfgp_comm_init(self._h, buf, int(rank), int(size)), "mfgp_comm_init")
        self.comm_rank, self.comm_size = int(rank), int(size)

    def comm_destroy(self):
        self._check(self._lib.mfgp_comm_destroy(self._h), "mfgp_comm_destroy")
        self.comm_rank, self.comm_size = 0, 1

    @property
    def comm_aborted(self):
        """the handle's communicator was torn down after a failed / unmatched collective: no further collective may be issued on it
        and the process should end with an error (mfgp_comm_state == -1)"""
        return self._h is not None and int(self._lib.mfgp_comm_state(self._h)) < 0

    def comm_calibrate(self, reps=20, panel_blocks=0):
        """collective: measure what a small collective of this handle's communicator costs (mfgp_comm_calibrate) -> dict; the figure
        stays on the handle and decides whether a shared evaluation's Cholesky is distributed over the group"""
        out = np.zeros(6)
        self._check(self._lib.mfgp_comm_calibrate(self._h, int(reps), int(panel_blocks), _dptr(out)), "mfgp_comm_calibrate")
        return {"broadcast_us": float(out[0]), "allgather_us": float(out[1]), "allgather_bytes_per_rank": int(out[2]),
                "allgather_GBps": float(out[3]), "reps": int(out[4]), "own_worst_median_us": float(out[5])}

    def shard_decision(self):
        """-> dict: how a shared evaluation of the current matrix will be planned (mfgp_shard_decision)"""
        out = np.zeros(6)
        self._check(self._lib.mfgp_shard_decision(self._h, _dptr(out)), "mfgp_shard_decision")
        return {"cholesky": "distributed" if out[0] else "replicated", "projected_saving_ms": float(out[1]), "collective_cost_ms": float(out[2]),
                "collectives_on_chain": int(out[3]), "measured_us_per_collective": float(out[4]), "forced_by_MFGP_DIST_CHOL": bool(out[5]),
                "why": ("MFGP_DIST_CHOL set" if out[5] else
                        ("no calibration: replicated until a collective of this group has been measured" if out[4] <= 0 else
                         "%d collectives x %.1f us = %.2f ms against %.2f ms of Cholesky flops saved (taken at saving > 1.25 x cost)"
                         % (int(out[3]), out[4], out[2], out[1])))}

    def dbg_fail_sharded_after(self, n):
        self._check(self._lib.mfgp_dbg_fail_sharded_after(self._h, int(n)), "mfgp_dbg_fail_sharded_after")

    def dbg_fail_collective_after(self, n):
        self._check(self._lib.mfgp_dbg_fail_collective_after(self._h, int(n)), "mfgp_dbg_fail_collective_after")

    def allgather_rows(self):
        """RCCL all-gather of the ranks' 128-row blocks of the device matrix (after kbuild_owned_rows with the communicator's rank and
        size): the lower part of every block, packed by owner -- half the bytes of full rows"""
        self._check(self._lib.mfgp_allgather_rows(self._h), "mfgp_allgather_rows")

    def allgather_host(self, send):
        """-> (size, count) array: row r = rank r's `send` (equal counts on every rank), through RCCL"""
        send = _c64(send).reshape(-1)
        out = np.empty((self.comm_size, send.shape[0]))
        self._check(self._lib.mfgp_allgather_host(self._h, _dptr(send), send.shape[0], _dptr(out)), "mfgp_allgather_host")
        return out

    def rows_download(self, row_begin, row_end):
        _, npad = self.dev_matrix()
        out = np.empty((int(row_end) - int(row_begin), npad))
        self._check(self._lib.mfgp_rows_download(self._h, int(row_begin), int(row_end), _dptr(out)), "mfgp_rows_download")
        return out

    def rows_upload(self, row_begin, block):
        block = _c64(block)
        self._check(self._lib.mfgp_rows_upload(self._h, int(row_begin), int(row_begin) + block.shape[0], _dptr(block)),
                    "mfgp_rows_upload")

    def eval_prebuilt(self, want_grad=True):
        nlml = ctypes.c_double()
        grad = np.zeros(self.n_params + 1)
        self._check(self._lib.mfgp_eval_prebuilt(self._h, int(bool(want_grad)), ctypes.byref(nlml), _dptr(grad)),
                    "mfgp_eval_prebuilt")
        return (nlml.value, grad) if want_grad else nlml.value

    def factorize(self, theta, noise, jitter=1e-8):
        theta = self._theta(theta)
        self._check(self._lib.mfgp_factorize(self._h, _dptr(theta), float(noise), float(jitter)), "mfgp_factorize")

    def nlml(self):
        v = ctypes.c_double()
        self._check(self._lib.mfgp_nlml(self._h, ctypes.byref(v)), "mfgp_nlml")
        return v.value

    def nlml_grad(self):
        g = np.zeros(self.n_params + 1)
        self._check(self._lib.mfgp_nlml_grad(self._h, _dptr(g)), "mfgp_nlml_grad")
        return g

    def append_row(self, x_new, y_new):
        """rank-1 append at the current hyper-parameters; True = appended, False = no padding slot left (refit needed)"""
        x = _c64(x_new).reshape(-1)
        if x.shape[0] != self.d:
            raise ValueError("x_new must have %d entries" % self.d)
        rc = self._lib.mfgp_append_row(self._h, _dptr(x), float(y_new))
        if rc == 0:
            self.n += 1
            return True
        if rc == 1:
            return False
        self._check(rc, "mfgp_append_row")

    def predict(self, Xstar, want_var=True, include_noise=True):
        Xs = _c64(Xstar)
        if Xs.ndim != 2 or Xs.shape[1] != self.d:
            raise ValueError("Xstar must be (N*, %d)" % self.d)
        mean = np.empty(Xs.shape[0])
        var = np.empty(Xs.shape[0]) if want_var else None
        rc = self._lib.mfgp_predict(self._h, _dptr(Xs), Xs.shape[0], _dptr(mean),
                                    _dptr(var) if want_var else None, int(bool(want_var)), int(bool(include_noise)))
        self._check(rc, "mfgp_predict")
        return mean, var

    # -- device-resident level chaining ------------------------------------------
    def augment(self, X, offsets):
        """this (low-fidelity) level's posterior mean on the stencil X + offsets[j] -> [X | means] (N, d + c)"""
        Xc, offs = _c64(X), _c64(offsets)
        if Xc.ndim != 2 or Xc.shape[1] != self.d or offs.ndim != 2 or offs.shape[1] != self.d:
            raise ValueError("X must be (N, %d) and offsets (c, %d)" % (self.d, self.d))
        out = np.empty((Xc.shape[0], self.d + offs.shape[0]))
        self._check(self._lib.mfgp_augment(self._h, _dptr(Xc), Xc.shape[0], _dptr(offs), offs.shape[0], _dptr(out)),
                    "mfgp_augment")
        return out

    def predict_chained(self, lf, Xstar, offsets, want_var=True, include_noise=True, want_aug=False):
        """predict of this level at [X* | lf means on the stencil]; the means stay on the device"""
        Xs, offs = _c64(Xstar), _c64(offsets)
        c = offs.shape[0] if offs.ndim == 2 else -1
        if Xs.ndim != 2 or offs.ndim != 2 or Xs.shape[1] != lf.d or offs.shape[1] != lf.d or self.d != lf.d + c:
            raise ValueError("Xstar must be (N*, d_lf), offsets (c, d_lf) and this level must have d_lf + c columns")
        mean = np.empty(Xs.shape[0])
        var = np.empty(Xs.shape[0]) if want_var else None
        aug = np.empty((Xs.shape[0], self.d)) if want_aug else None
        rc = self._lib.mfgp_predict_chained(self._h, lf._h, _dptr(Xs), Xs.shape[0], _dptr(offs), c, _dptr(mean),
                                            _dptr(var) if want_var else None, int(bool(want_var)),
                                            int(bool(include_noise)), _dptr(aug) if want_aug else None)
        self._check(rc, "mfgp_predict_chained")
        return (mean, var, aug) if want_aug else (mean, var)

    # -- read-back ------------------------------------------------------------------------------
    def _get_mat(self, fn, who):
        out = np.empty((self.n, self.n))
        self._check(fn(self._h, _dptr(out)), who)
        return out

    def get_K(self):
        return self._get_mat(self._lib.mfgp_get_K, "mfgp_get_K")

    def get_L(self):
        return self._get_mat(self._lib.mfgp_get_L, "mfgp_get_L")

    def get_Linv(self):
        return self._get_mat(self._lib.mfgp_get_Linv, "mfgp_get_Linv")

    def get_Kinv(self):
        return self._get_mat(self._lib.mfgp_get_Kinv, "mfgp_get_Kinv")

    def get_alpha(self):
        out = np.empty(self.n)
        self._check(self._lib.mfgp_get_alpha(self._h, _dptr(out)), "mfgp_get_alpha")
        return out

    def timings(self):
        t = Timings()
        self._check(self._lib.mfgp_get_timings(self._h, ctypes.byref(t)), "mfgp_get_timings")
        return t.as_dict()

    def device_synchronize(self):
        """hipDeviceSynchronize on this engine's device (benchmarks bracket their timed region with it)"""
        self._check(self._lib.mfgp_device_synchronize(self._h), "mfgp_device_synchronize")

    def counters(self, reset=False):
        c = Counters()
        self._check(self._lib.mfgp_get_counters(self._h, ctypes.byref(c), int(bool(reset))), "mfgp_get_counters")
        return c.as_dict()

    # -- kernel-level test hooks ----------------------------------------------------------------
    def dbg_gemm_nt(self, A, B, C, alpha=1.0, beta=0.0, tile=128):
        A, B = _c64(A), _c64(B)
        C = np.array(C, dtype=np.float64, order="C", copy=True)
        M, K = A.shape
        N = B.shape[0]
        self._check(self._lib.mfgp_dbg_gemm_nt(self._h, _dptr(A), _dptr(B), _dptr(C), M, N, K, float(alpha),
                                               float(beta), int(tile)), "mfgp_dbg_gemm_nt")
        return C

    def dbg_leaf(self, A):
        A = _c64(A)
        L = np.empty((128, 128))
        X = np.empty((128, 128))
        ld = ctypes.c_double()
        rc = self._lib.mfgp_dbg_leaf(self._h, _dptr(A), _dptr(L), _dptr(X), ctypes.byref(ld))
        if rc < 0:
            self._check(rc, "mfgp_dbg_leaf")
        return L, X, ld.value, rc
