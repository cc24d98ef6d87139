#!/usr/bin/env python3
"""bench.py -- the north-star measurement: fit + predict wall-time of a 2-fidelity NARGP at
N_lf = N_hf = N* = 8192, d = 4, fp64 (BASELINE.json `metric`; SURVEY.md 8(d) "North-star").

One STEP = one complete pass of the hot path over one batch of synthetic input:
  level 1 (low fidelity):  GPRegression(X_lf, Y_lf), RBF(4)           -> optimize()            1 L-BFGS-B run
  level 2 (high fidelity): X_aug = [X_hf | mean_1(X_hf)], k1*k2+k3     -> ARD recipe            1 + 6 runs
  predict:                 X*_aug = [X* | mean_1(X*)]                  -> mean, variance at N* points
following src/MFDataFusion.py:75-100,141-156 and src/abstractMFGP.py:82-106,131-137 of the reference.
Optimiser trajectories are not comparable across back-ends, so every L-BFGS-B run gets the same FIXED
objective+gradient evaluation budget (--evals, default 20; SURVEY.md 8(d) "fixed-budget fit"), enforced EXACTLY
(`eval_cap`): with scipy's `maxfun` alone a run overshoots by up to a line search, and the total then moves by
+-8 % with the last bits of the arithmetic (173 vs 188 evaluations between two versions of one kernel).

  python bench.py [--gpus N --steps K --warmup W]

N > 1 is one process per GPU.  Under a launcher (torch.distributed.run exports RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_*) this process IS one rank.  Started as a plain process with --gpus N > 1 it is the launcher: before anything
touches HIP it starts N copies of itself, one per GPU, with that environment (and a per-job token for the
rendezvous), waits for them, and exits non-zero if any rank did.  A rank whose RCCL communicator does not come up
prints the reason and exits non-zero -- an N-GPU line is never produced by a job that fell back to TCP
(`--single-device` is the explicit rehearsal of the N-rank code path on one GPU, RCCL included: the ranks pose as
separate hosts -- sharding.rehearsal_env -- so that RCCL builds a real N-rank communicator over its socket transport;
its line says n_gpus = 1).

N > 1 is STRONG scaling of the same job: the randomized restarts and the N* predictive rows are sharded over the
ranks (sharding.py: rendezvous + tiny object gathers over TCP, device collectives = RCCL inside libmfgp_hip.so; no
PyTorch anywhere); rank 0 alone runs the sequential first HF run -> restart 0.
Prints ONE JSON line on rank 0.  `value` = milliseconds per fit+predict (lower is better).
The CPU comparator (`cpu_baseline`) is the numpy/LAPACK oracle timed on the host cores of the same box
on a bounded sample (one objective+gradient evaluation per level + the predict products), scaled by the
number of evaluations the GPU path actually issued.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X vendor fp64 peak (matrix = vector); the guide lists no fp64 row
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def hf_4d(x):
    return (np.prod(np.sin(np.pi * x[:, :4]), axis=1) + 5.0)[:, None]


def lf_4d(x):
    return hf_4d(x) - 0.25 * (np.sin(x[:, 0] * np.pi * 0.1) + np.sin(x[:, 1] * np.pi * 0.05)
                              + np.sin(x[:, 2] * 0.15 * np.pi) + np.sin(x[:, 3] * 0.2 * np.pi))[:, None]


def make_data(n_lf, n_hf, n_star, seed=2):
    rng = np.random.default_rng(seed)
    X_lf = rng.uniform(size=(n_lf, 4))
    X_hf = rng.uniform(size=(n_hf, 4))
    X_st = rng.uniform(size=(n_star, 4))
    return X_lf, lf_4d(X_lf), X_hf, X_st


def one_step(args, comm, engines, data):
    """one fit+predict of the 2-fidelity NARGP; returns (mean, var, model)"""
    from multifidelity_datafusion_gps_amd import NARGP
    X_lf, Y_lf, X_hf, X_st = data

    class BudgetNARGP(NARGP):
        lf_max_iters = args.evals
        first_run_max_iters = args.evals
        restart_max_iters = args.evals
        num_restarts = args.restarts
        restart_concurrency = args.concurrency
        restart_lockstep = None if args.lockstep < 0 else bool(args.lockstep)
        lockstep_lanes = args.lanes if args.lanes > 0 else None
        lockstep_width = args.width if args.width > 0 else None
        restart_aux = args.aux if args.aux > 0 else None
        restart_lend_main = bool(args.lend_main)
        eval_cap = args.evals          # exact: scipy's maxfun alone lets a run overshoot by a line search

    t0 = time.perf_counter()
    model = BudgetNARGP(4, f_exact=hf_4d, f_low=None, lf_X=X_lf, lf_Y=Y_lf, seed=args.seed, comm=comm,
                        engines=engines)
    t1 = time.perf_counter()
    model.fit(X_hf)
    t2 = time.perf_counter()
    mean, var = model.predict(X_st)
    t3 = time.perf_counter()
    # the sequential pieces of the job: the LF run, rank 0's first HF run -> restart 0, the predict
    chain_evals = sum(r.n_evals for r in model.hf_model.optimization_runs if not r.background)
    model.phase = {"lf_ms": (t1 - t0) * 1e3, "fit_ms": (t2 - t1) * 1e3, "predict_ms": (t3 - t2) * 1e3, "chain_evals": chain_evals}
    # what the timed workload's own fit did: a jitter retry is a whole extra evaluation -- it would silently change
    # the evaluation count the value is quoted on -- and the fitted noise variances say how ill-conditioned the timed factorisations are
    lf_m = getattr(model, "lf_model", None)
    model.phase["jitter_retries"] = int(model.hf_model.n_jitter_retries + (lf_m.n_jitter_retries if lf_m is not None else 0))
    model.phase["failed_evaluations"] = int(model.hf_model.n_failed_evals + (lf_m.n_failed_evals if lf_m is not None else 0))
    model.phase["fitted_noise_variance"] = {"hf": float(model.hf_model.likelihood.variance.value),
                                            "lf": float(lf_m.likelihood.variance.value) if lf_m is not None else None}
    model.phase["fit_driver"] = model.last_fit_info
    lanes = getattr(model, "last_lockstep_lanes", None)
    if lanes:
        model.phase["lockstep"] = [{"rounds": ls.rounds, "evaluations": ls.evals, "engine_ms": round(ls.engine_s * 1e3, 1),
                                    "largest_round": max(ls.round_sizes) if ls.round_sizes else 0} for ls in lanes]
    return mean, var, model


def _blas_info():
    try:
        from threadpoolctl import threadpool_info
        infos = [p for p in threadpool_info() if p.get("user_api") == "blas"]
        if infos:
            p = max(infos, key=lambda q: q.get("num_threads", 1))
            return {"vendor": p.get("internal_api"), "version": p.get("version"), "threads": int(p.get("num_threads", 1)),
                    "threading_layer": p.get("threading_layer")}
    except Exception:  # noqa: BLE001
        pass
    return {"vendor": "unknown", "version": None, "threads": os.cpu_count() or 1, "threading_layer": None}


def _host_cpu():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def _pick_blas_threads():
    """OpenBLAS on a box whose CPU share is smaller than its core count (16 of 256 visible) is slowest at the default of one
    thread per visible core: time dpotrf + dpotri at N = 4096 under 8 / 16 / 32 / all threads once and keep the best.
    Same protocol as _lapack_share (Fortran-ordered operand, one untimed call per setting, best of 2, info checked), so that the
    rates of the two are comparable."""
    try:
        from scipy.linalg import lapack
        from threadpoolctl import threadpool_limits
    except Exception:  # noqa: BLE001
        return None, {}
    rng = np.random.default_rng(0)
    n = 4096
    M = rng.standard_normal((n, n))
    A = np.asfortranarray(M.dot(M.T) + n * np.eye(n))
    ncpu = os.cpu_count() or 1
    tried = {}
    detail = {}
    for t in sorted({min(t, ncpu) for t in (8, 16, 32, ncpu)}):
        with threadpool_limits(limits=t, user_api="blas"):
            best, b1, b2 = np.inf, np.inf, np.inf
            for rep in range(3):
                t0 = time.perf_counter()
                L, i1 = lapack.dpotrf(A, lower=1, overwrite_a=0)
                t1 = time.perf_counter()
                _, i2 = lapack.dpotri(L, lower=1, overwrite_c=0)
                t2 = time.perf_counter()
                if rep > 0 and i1 == 0 and i2 == 0:
                    best, b1, b2 = min(best, t2 - t0), min(b1, t1 - t0), min(b2, t2 - t1)
        tried[t] = round(n ** 3 / best / 1e9, 1)      # dpotrf n^3/3 + dpotri 2 n^3/3
        # the two routines run at very different rates under OpenBLAS (dpotrf's panel recursion threads poorly, dpotri is
        # dtrmm-like): both are reported so that the rates of _lapack_share at the full size can be held against them
        detail[t] = {"dpotrf_gflops": round(n ** 3 / 3 / b1 / 1e9, 1), "dpotri_gflops": round(2 * n ** 3 / 3 / b2 / 1e9, 1)}
    _pick_blas_threads.detail = detail
    return max(tried, key=tried.get), tried


def _lapack_share(Ky, reps=2):
    """dpotrf / dtrtri / dpotri (the O(N^3) part of one evaluation) on Ky, each on a FORTRAN-ordered operand prepared outside the
    timer (f2py otherwise copies a C-ordered N^2 array inside the call), after one untimed call (thread pool, page faults), best
    of `reps`, `info` checked -> seconds and GFLOP/s per routine"""
    from scipy.linalg import lapack
    n = float(Ky.shape[0])
    Kf = np.asfortranarray(Ky)
    L, info = lapack.dpotrf(Kf, lower=1, clean=1, overwrite_a=0)        # (also the warm-up)
    if info != 0:
        return {"error": "dpotrf info = %d" % info}
    L = np.asfortranarray(L)
    out, total = {}, 0.0
    for name, fn, flops in (("dpotrf", lambda: lapack.dpotrf(Kf, lower=1, clean=1, overwrite_a=0), n ** 3 / 3),
                            ("dtrtri", lambda: lapack.dtrtri(L, lower=1, overwrite_c=0), n ** 3 / 3),
                            ("dpotri", lambda: lapack.dpotri(L, lower=1, overwrite_c=0), 2 * n ** 3 / 3)):
        best = np.inf
        for rep in range(reps + (0 if name == "dpotrf" else 1)):         # dtrtri / dpotri: their first call is the warm-up
            t0 = time.perf_counter()
            res = fn()
            dt = time.perf_counter() - t0
            if res[-1] != 0:
                return {"error": "%s info = %d" % (name, res[-1])}
            if name == "dpotrf" or rep > 0:
                best = min(best, dt)
        out[name + "_s"] = round(best, 3)
        out[name + "_gflops"] = round(flops / best / 1e9, 1)
        total += best
    out["gflops"] = round(n ** 3 * (4.0 / 3) / total / 1e9, 1)
    out["note"] = "Fortran-ordered operands, one untimed call first, best of %d, info == 0 checked" % reps
    return out


def cpu_baseline(args, data, n_lf_evals, n_hf_evals, budget_s=170.0):
    """the oracle (numpy + LAPACK dpotrf / dtrtri / dpotri / dpotrs = the routines GPy calls) on the host cores of this
    box, as BASELINE.md section 2 words it: per level ONE WARM-UP evaluation, then the best of (up to) three timed
    objective+gradient evaluations at full size, + the predict products -- a bounded sample, EXTRAPOLATED to the evaluations
    the GPU run issued (`kind` says so; the measured and the extrapolated seconds are separate fields)."""
    from oracle import gp_oracle as orc
    X_lf, Y_lf, X_hf, X_st = data
    threads, tried = _pick_blas_threads()
    limiter = None
    if threads:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads, user_api="blas")
    blas = _blas_info()
    t_start = time.perf_counter()

    def timed_evals(parts, th, nz, X, Y, deadline):
        """-> (timed seconds per evaluation, state): one warm-up, then at least one and at most three timed evaluations (the
        second and third only while the deadline allows)"""
        st = orc.inference(parts, th, nz, X, Y, want_grad=True)             # warm-up: never timed
        ts = []
        while len(ts) < 3:
            t0 = time.perf_counter()
            st = orc.inference(parts, th, nz, X, Y, want_grad=True)
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() + ts[-1] > deadline:
                break
        return ts, st

    parts_lf = [(orc.RBF, 0, 4, 0)]
    th_lf, nz_lf = np.array([1.0, 1.0]), 1.0
    lf_ts, st_lf = timed_evals(parts_lf, th_lf, nz_lf, X_lf, Y_lf, t_start + 0.45 * budget_s)
    t0 = time.perf_counter()
    aug_hf = orc.cov(parts_lf, th_lf, X_lf, X_hf).T.dot(st_lf["alpha"])      # LF posterior mean at X_hf and X* (mean only)
    aug_st = orc.cov(parts_lf, th_lf, X_lf, X_st).T.dot(st_lf["alpha"])
    lf_means_s = time.perf_counter() - t0
    Xa = np.hstack([X_hf, aug_hf[:, None]])
    Xsa = np.hstack([X_st, aug_st[:, None]])
    parts_hf = [(orc.RBF, 4, 5, 0), (orc.RBF, 0, 4, 0), (orc.RBF, 0, 4, 1)]
    th_hf = np.ones(6)
    Y_hf = hf_4d(X_hf)
    nz_hf = 0.01 * Y_hf.var()
    hf_ts, st_hf = timed_evals(parts_hf, th_hf, nz_hf, Xa, Y_hf, t_start + 0.9 * budget_s)
    t0 = time.perf_counter()
    orc.predict(parts_hf, th_hf, nz_hf, Xa, st_hf, Xsa)
    hf_predict_s = time.perf_counter() - t0
    # how much of one evaluation is LAPACK (the O(N^3) part) and at what rate this host runs it
    lap = _lapack_share(st_hf["K"] + (nz_hf + 1e-8) * np.eye(len(Xa)))
    measured = time.perf_counter() - t_start
    if limiter is not None:
        limiter.restore_original_limits()
    lf_eval_s, hf_eval_s = min(lf_ts), min(hf_ts)
    if "error" not in lap:
        lap["seconds_of_one_hf_eval"] = round(lap["dpotrf_s"] + lap["dtrtri_s"] + lap["dpotri_s"], 3)
        lap["share_of_one_hf_eval"] = round(lap["seconds_of_one_hf_eval"] / hf_eval_s, 3)
    total_s = n_lf_evals * lf_eval_s + n_hf_evals * hf_eval_s + lf_means_s + hf_predict_s
    # ... and the same evaluations priced at the LAPACK routines alone (both levels factorise an N x N matrix): what a host code
    # without numpy's N^2 temporaries could not go below
    lapack_only_ms = (round((n_lf_evals + n_hf_evals) * lap["seconds_of_one_hf_eval"] * 1e3, 1) if "error" not in lap else None)
    return {"value": round(total_s * 1e3, 1), "unit": "ms", "cores": int(blas["threads"]), "kind": "port-extrapolated",
            "lapack_only_ms": lapack_only_ms,
            "lapack_only_is": "(LF + HF evaluations the GPU run issued) x the measured dpotrf + dtrtri + dpotri seconds of one N x N evaluation",
            "blas": blas, "blas_threads_tried_gflops": tried,
            "blas_threads_tried_per_routine_n4096": getattr(_pick_blas_threads, "detail", None), "host_cpu": _host_cpu(),
            "measured_s": round(measured, 2),
            "extrapolated_s": round(total_s, 1),
            "lf_eval_s": [round(t, 2) for t in lf_ts], "hf_eval_s": [round(t, 2) for t in hf_ts],
            "warmed_up": {"lf": True, "hf": True},
            "lf_means_s": round(lf_means_s, 2), "hf_predict_s": round(hf_predict_s, 2), "lapack_share_of_one_hf_eval": lap,
            "sample": "oracle (numpy + LAPACK, the GPy algorithm) at full size on this host with %d BLAS threads (best of %s "
                      "GFLOP/s on a 4096^2 dpotrf + dpotri): per level one untimed warm-up evaluation, then %d LF and %d HF "
                      "objective+gradient evaluations timed after a warm-up (best %.2f s / %.2f s), LF means %.2f s, HF predict "
                      "%.2f s = %.1f s measured; value = those times EXTRAPOLATED to the %d LF + %d HF evaluations the GPU run "
                      "issued + the predicts"
                      % (int(blas["threads"]), tried, len(lf_ts), len(hf_ts), lf_eval_s, hf_eval_s, lf_means_s, hf_predict_s,
                         measured, n_lf_evals, n_hf_evals)}


PMC_FILE = os.path.join("profiles", "r06_pmc.json")
MFMA_FILE = os.path.join("profiles", "r06_mfma_counters.json")
BARE_MFMA_TFLOPS = 71.0   # bare v_mfma_f64_4x4x4_4b loop on this part (profiles/r03_probes.txt): what the instruction itself can issue


def committed_counters(path, build_id):
    """-> (dict, None) for a committed counter summary taken with THIS library (its `csrc_hash` equals the id embedded in the
    loaded libmfgp_hip.so), else (None, reason): counters of another build are not reported as this run's"""
    try:
        d = json.load(open(os.path.join(ROOT, path)))
    except Exception as ex:  # noqa: BLE001
        return None, "%s: %s" % (path, ex.__class__.__name__)
    have = d.get("csrc_hash")
    if not have:
        return None, "%s carries no csrc_hash (taken before the sources were stamped)" % path
    if have != build_id:
        return None, "%s was taken with library build %s, this run loaded %s" % (path, have, build_id)
    return d, None


def pmc_traffic(kernel, n, build_id):
    """-> (HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes, or None; the reason when None)"""
    d, why = committed_counters(PMC_FILE, build_id)
    if d is None:
        return None, why
    if int(d.get("n", -1)) != int(n):
        return None, "%s was taken at N = %s" % (PMC_FILE, d.get("n"))
    try:
        return int(d[kernel]["traffic_bytes"]), None
    except Exception:  # noqa: BLE001
        return None, "%s has no entry for %s" % (PMC_FILE, kernel)


def mfma_busy(build_id):
    """-> (matrix-pipe busy fraction per kernel from the committed SQ counter pass over one evaluation at N = 8192
    (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); profiles/README.md has the calibration), or None; the reason)"""
    d, why = committed_counters(MFMA_FILE, build_id)
    if d is None:
        return None, why
    try:
        return {k: v["mfma_busy"] for k, v in d["pmcA_eval"].items() if v.get("mfma_busy", 0) > 0}, None
    except Exception:  # noqa: BLE001
        return None, "%s has no pmcA_eval block" % MFMA_FILE


def _rate(num, ms):
    return num / (ms * 1e-3) if ms > 0 else 0.0


_PROFILER_ENV = ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")


def under_profiler(env=None):
    """is this process running under rocprofv3 / rocprof (a preloaded profiler tool library)?  (An LD_PRELOAD as such is not the
    sign: the GPU pool's own harness preloads an exec guard into every process.)"""
    env = os.environ if env is None else env
    preload = env.get("LD_PRELOAD", "").lower()
    return (any(w in preload for w in ("rocprof", "roctracer", "rocp_")) or bool(env.get("ROCP_TOOL_LIBRARIES"))
            or bool(env.get("HSA_TOOLS_LIB")) or any(k.startswith("ROCPROFILER_") or k.startswith("ROCPROF_") for k in env))


def start_power_watch():
    """--power: socket power / shader clock sampled by a CHILD process (tools/power_watch.py --until-eof) beside the run.
    OPT-IN: the sampler costs host CPU beside the L-BFGS-B driver threads, so the default bench line is taken without
    it.  Never under a profiler: the child tree would inherit the profiler's preloaded library, and every exec in it is the
    exec-after-GPU-init this pool forbids -- returns None there; the child's environment is stripped of the preload variables
    in any case, and the sampler reads sysfs (no process per sample) where the driver exposes the sensors."""
    import subprocess
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "power_watch.py")
    if not os.path.exists(tool) or under_profiler():
        return None
    env = {k: v for k, v in os.environ.items()
           if k not in ("ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB") and not k.startswith("ROCPROFILER_") and not k.startswith("ROCPROF_")}
    try:
        return subprocess.Popen([sys.executable, tool, "--period", "0.1", "--until-eof"], stdin=subprocess.PIPE,
                                stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)
    except OSError:
        return None


def stop_power_watch(proc, wall0, wall1):
    """-> summary of the samples that fall into the timed region [wall0, wall1] (unix seconds), or None"""
    try:
        out, _ = proc.communicate(input="", timeout=20)
        d = json.loads(out.strip().splitlines()[-1])
    except Exception:  # noqa: BLE001 - diagnostic only
        try:
            proc.kill()
        except OSError:
            pass
        return None
    rows = [r for r in d.get("samples", []) if wall0 <= r[0] <= wall1]
    if not rows:
        return None
    med = lambda v: sorted(v)[len(v) // 2]   # noqa: E731
    pw, ck = [r[1] for r in rows], [r[2] for r in rows]
    hot = [r for r in rows if d.get("cap_w") and r[1] >= 0.93 * d["cap_w"]]
    return {"source": d.get("source", "rocm-smi") + " sampled every ~0.1 s by a child process over the timed region (tools/power_watch.py)",
            "cap_w": d.get("cap_w"), "samples": len(rows), "socket_w_median": med(pw), "socket_w_max": max(pw),
            "sclk_mhz_median": med(ck), "sclk_mhz_min": min(ck), "junction_c_max": max(r[3] for r in rows),
            "fraction_of_samples_within_7pct_of_cap": round(len(hot) / len(rows), 3),
            "note": "with the concurrent restarts in flight the socket sits at its power cap and the clock is managed down; "
                    "a bare v_mfma_f64_4x4x4_4b loop holds 71 TFLOP/s at 2.40 GHz and ~1130 W (profiles/r03_sustained_mfma_probe.json)"}


EXIT_JITTER_RETRIES = 5   # --strict-jitter: the timed steps repeated an evaluation with more jitter
EXIT_PEER_LOST = 4    # a rank whose peer vanished mid-collective (the launcher reports the rank that vanished, not this one)


def sharding_note(world, restarts, rccl_ranks):
    """what an N-rank job shards, in words (config.sharding of the JSON line)"""
    if world == 1:
        return "one rank: nothing is sharded"
    from multifidelity_datafusion_gps_amd.abstractMFGP import AbstractMFGP
    assign = AbstractMFGP.assign_restarts(restarts, world)
    chain = [0] + [r for r in range(1, world) if not assign[r]]
    shared = rccl_ranks == world
    return ("predictive rows by contiguous blocks over the ranks; the randomized restarts dealt to the ranks %s (each rank runs its own in lock step "
            "over batched evaluations); the LF run: one optimiser on rank 0, %s; the sequential pair first HF run -> restart 0: rank 0%s"
            % (assign, "every evaluation shared by all ranks (mfgp_eval_sharded: Cholesky replicated, rows of L^-T / K^-1 by 128-row block, one "
                       "all-gather + one all-reduce)" if shared else "the others adopt its optimum",
               (", its evaluations shared with ranks %s, which were dealt no restart" % chain[1:]) if (shared and len(chain) > 1 and not assign[0])
               else " alone (every other rank has restarts of its own)"))


def launch_ranks(n_ranks, argv, script=None):
    """the --gpus N > 1 job started as a plain process: N child processes of this script, one rank per GPU.  Runs BEFORE
    this process imports the engine or touches HIP (it never does).  Children inherit stdout / stderr, so rank 0's JSON
    line is this job's line; returns the exit code of the job (the first non-zero child code, else 0)."""
    import secrets
    import socket
    import subprocess
    # two ports probed free: MASTER_PORT for whatever store a rank's own launcher-side code may open, MFGP_COMM_PORT for the
    # rendezvous hub of sharding.comm_from_env (which would otherwise derive MASTER_PORT +- 1000 -- a port nobody checked)
    with socket.socket() as sk, socket.socket() as sk2:
        sk.bind(("127.0.0.1", 0))
        sk2.bind(("127.0.0.1", 0))
        port, comm_port = sk.getsockname()[1], sk2.getsockname()[1]
    base = dict(os.environ)
    base.update({"WORLD_SIZE": str(n_ranks), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                 "MFGP_COMM_PORT": str(comm_port), "MFGP_COMM_TOKEN": secrets.token_hex(16)})
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts (RCCL across processes)
    procs = []
    for r in range(n_ranks):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env))
    failed = []                         # (rank, code) in the order the launcher saw them end
    live = list(procs)
    deadline = None
    while live:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                failed.append((procs.index(p), code))
        if failed and deadline is None:
            # a rank that fails takes its peers down with it (their next collective raises ConnectionError -> EXIT_PEER_LOST):
            # give them a moment to say so, so that the code reported is the ORIGINATING rank's, then stop whatever is left
            deadline = time.monotonic() + 1.0
        if deadline is not None and time.monotonic() >= deadline:
            for q in live:              # exactly the processes started above: they would wait for ever in a collective
                q.terminate()
            deadline = float("inf")
    rc = 0
    if failed:
        origin = [f for f in failed if f[1] != EXIT_PEER_LOST] or failed
        r0, c0 = origin[0]
        rc = c0 if c0 > 0 else 128 - c0
        sys.stderr.write("bench.py: rank %d exited with code %d; the other ranks were stopped%s\n" % (
            r0, c0, "".join(" [rank %d: %d]" % f for f in failed if f != (r0, c0))))
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", dest="n", type=int, default=8192, help="N_lf = N_hf = N*")
    ap.add_argument("--evals", type=int, default=20, help="objective evaluations per L-BFGS-B run")
    ap.add_argument("--restarts", type=int, default=6)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--lockstep", type=int, default=-1,
                    help="-1 (default) and 1: the package's default -- the HF level's 1 + 6 runs as lock-stepped runs over batched "
                         "evaluations (mfgp_eval_batch); 0: round 3's concurrent restarts on auxiliary handles (--concurrency)")
    ap.add_argument("--lanes", type=int, default=0, help="engine handles the lock-stepped runs are dealt to (0: the model's default: 1 at N >= 6144, else 2)")
    ap.add_argument("--width", type=int, default=0, help="live lock-step slots per rank (0: half the rank's runs, rounded up)")
    ap.add_argument("--concurrency", type=int, default=2,
                    help="--lockstep 0 only: randomized restarts in flight beside the main run (auxiliary engine handles per rank)")
    ap.add_argument("--aux", type=int, default=0, help="auxiliary engine handles of the concurrent restarts (0: --concurrency of them)")
    ap.add_argument("--aux-order", default="natural", choices=["natural", "reversed"])
    ap.add_argument("--lend-main", type=int, default=0, help="1: the main engine joins the restarts' pool after restart 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip `extra_configs` (cfg2 / cfg3 / cfg5 of BASELINE.json measured after the timed region, outside `value`: bench_extra.py, ~10 s)")
    ap.add_argument("--power", action="store_true", help="sample socket power / sclk beside the timed region (a child process; "
                                                         "off by default: it costs host CPU beside the optimiser threads; never under a profiler)")
    ap.add_argument("--no-power", action="store_true", help="(accepted for old command lines; power sampling is off unless --power)")
    ap.add_argument("--strict-jitter", action="store_true",
                    help="exit 5 without a line if a timed step repeated an evaluation with more jitter, or if the steps differ in it "
                         "(default: the retries are counted in the line -- they are part of the workload as GPy's jitchol would run it)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses GPU 0; RCCL is made to accept that by giving every rank its own "
                         "NCCL_HOSTID (a real N-rank communicator over RCCL's socket transport on loopback)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))     # nothing below runs in the launcher; no HIP call was made
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != max(args.gpus, 1):
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks\n" % (args.gpus, world))
        sys.exit(2)
    local_rank = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    # No PyTorch: the host side needs a rendezvous and a few tiny object gathers (sharding.SocketComm, TCP on
    # MASTER_ADDR), the device collectives are RCCL inside libmfgp_hip.so on the engine's own stream.
    # dmabuf IPC only on these hosts (RCCL across processes); under torch.distributed.run this process IS a rank and inherits
    # the caller's environment: make sure of it before the HIP runtime initialises
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    power_proc = None
    if rank == 0 and args.power and not args.no_power:
        power_proc = start_power_watch()       # a child that samples the sensors; started BEFORE this process touches HIP
    # HIP's stream -> hardware-queue mapping: 2 queues per priority measured best for this job (profiles/r03_hw_queues.txt).  The
    # package no longer changes it behind the host application's back: the bench asks for it explicitly, before the
    # HIP runtime initialises, and reports what was in effect.
    os.environ.setdefault("MFGP_HW_QUEUES", "2")
    from multifidelity_datafusion_gps_amd import sharding
    if world > 1 and args.single_device:
        os.environ.update(sharding.rehearsal_env(rank))      # before librccl is loaded (lazily, by attach_engine)
    from multifidelity_datafusion_gps_amd._lib import Engine
    comm = sharding.comm_from_env()
    try:
        first = Engine(local_rank)
    except RuntimeError as ex:
        # a launcher that narrows the visible devices per rank (HIP_VISIBLE_DEVICES = one GPU each) leaves only device 0
        if local_rank == 0 or "bad device id" not in str(ex):
            raise
        local_rank = 0
        first = Engine(0)
    engines = {"lf": first, "hf": Engine(local_rank)}
    # creation ORDER matters: HIP maps the handles' streams onto its hardware queues by creation, and handles that share a queue
    # serialise (tools/queue_pairs.py: of four handles created back to back, 0 + 3 and 1 + 2 share).  --aux-order reversed creates
    # hf#2 before hf#1, so that the first auxiliary handle lands on the lane the main handle (hf) is NOT on.
    lanes = args.lanes if args.lanes > 0 else (2 if 768 <= args.n < 6144 else 1)      # the model's by-size rule
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    lockstep = bool(args.lockstep) if args.lockstep >= 0 else True      # the model's default
    n_aux = (lanes - 1) if lockstep else max(args.concurrency, 2)
    js = list(range(1, n_aux + 1))
    for j in (reversed(js) if args.aux_order == "reversed" else js):
        engines["hf#%d" % j] = Engine(local_rank)
    collectives = "none (1 rank)"
    if world > 1:
        # librccl prints a version banner on the C-level stdout when its first communicator comes up: this job's stdout carries
        # ONE JSON line, so the banner goes to stderr
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            # the world communicator lives in the LOW-fidelity level's handle: its run is then shared by all ranks
            # (mfgp_eval_sharded) and the high-fidelity handle stays free for the group that shares first run -> restart 0
            comm.attach_engine(engines["lf"], required=True)
        except sharding.RcclInitError as ex:
            # agreed by every rank over TCP: all of them end here.  No fall-back to TCP: a line that says n_gpus = N
            # must come from N ranks on RCCL.  os._exit: where the initialisation hangs, its thread never returns.
            sys.stderr.write("bench.py: rank %d: the RCCL communicator of %d ranks was not created: %s\n" % (rank, world, ex))
            sys.stderr.flush()
            os._exit(3)
        finally:
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        collectives = "rccl (ncclAllGather on the engine stream; rendezvous + object gathers over tcp)"
        if args.single_device:
            collectives += "; single-device rehearsal: %d ranks on GPU 0 posing as %d hosts (NCCL_HOSTID), RCCL socket transport" % (world, world)

    def barrier():
        engines["hf"].device_synchronize()
        comm.barrier()
        engines["hf"].device_synchronize()

    data = make_data(args.n, args.n, args.n)
    for _ in range(args.warmup):
        one_step(args, comm, engines, data)
    for e in engines.values():
        e.counters(reset=True)
    barrier()
    t0 = time.perf_counter()
    wall0 = time.time()
    phases = []
    for _ in range(args.steps):
        mean, var, model = one_step(args, comm, engines, data)
        phases.append(model.phase)
    barrier()
    dt = max(comm.allgather_object(time.perf_counter() - t0))     # max over ranks
    # jitter retries (GPy's jitchol: an evaluation whose Ky is not positive definite is repeated with more jitter) are part of the
    # workload as the reference's engine would execute it -- the optimiser does visit such points at this budget (sigma_n^2 -> 1e-11) --
    # and each is a whole extra evaluation: they are COUNTED in the line, and they must be the same in every timed step (same seeds,
    # same arithmetic), or the evaluation count the value is quoted on would not be a property of the workload
    per_step = [p["jitter_retries"] for p in phases]
    retries = sum(comm.allgather_object(sum(per_step)))      # (collective: every rank takes the same exit)
    retries_stable = all(comm.allgather_object(len(set(per_step)) <= 1))
    if args.strict_jitter and (retries or not retries_stable):
        sys.stderr.write("bench.py: rank %d: --strict-jitter: the timed steps saw %d jitter retries (per step on this rank: %s)\n"
                         % (rank, retries, per_step))
        sys.stderr.flush()
        os._exit(EXIT_JITTER_RETRIES)
    wall1 = time.time()
    power = stop_power_watch(power_proc, wall0, wall1) if power_proc is not None else None
    ms_per_step = dt * 1e3 / args.steps

    # outside the timed region: the row-block K build + RCCL all-gather layout of north_star,
    # one evaluation each way on the HF level, reported next to the local build it competes with
    rowblock = None
    if world > 1:
        try:
            th, nz = np.ones(2), 0.05
            e = engines["lf"]                 # (the handle that carries the world communicator)
            e.eval(th, nz)
            barrier()
            t1 = time.perf_counter()
            f_loc = e.eval(th, nz, want_grad=False)
            barrier()
            t2 = time.perf_counter()
            f_rb = sharding.eval_rowblock_allgather(e, comm, th, nz, want_grad=False)
            barrier()
            t3 = time.perf_counter()
            rowblock = {"local_build_eval_ms": round((t2 - t1) * 1e3, 3), "rowblock_allgather_eval_ms": round((t3 - t2) * 1e3, 3),
                        "nlml_equal": bool(f_loc == f_rb), "transport": comm.transport}
        except Exception as ex:  # noqa: BLE001 - diagnostic only, never fails the bench line
            rowblock = {"error": repr(ex)[:200]}

    if rank == 0:
        clf = engines["lf"].counters()
        chf = engines["hf"].counters()
        for k_, e_ in engines.items():
            if k_.startswith("hf#"):
                for kk, vv in e_.counters().items():
                    chf[kk] += vv
        tot = {k: clf[k] + chf[k] for k in clf}
        evals = tot["evals"]
        # The LF level's evaluations run ALONE on the GPU (its single L-BFGS-B run is sequential): "uncontended".  The
        # HF level's share the GPU with the concurrent restarts, which stretches every launch but shortens the job.
        # Rate the GPU delivered on the sweeps over the timed region: several evaluations are in flight at once (the concurrent
        # restarts), so per-launch event times overlap; the aggregate = all sweep flops / timed wall time (conservative: the
        # wall time also holds the K builds, solves, gradient reductions and the predicts).
        # (with MFGP_KINV_STREAM=0 the K^-1 product is a stand-alone launch counted in kinv_flops: the same Np^3 per evaluation either way)
        sweep_tf = (tot["cholinv_flops"] + tot["kinv_flops"]) / dt / 1e12
        sweep_tf_launch = _rate(tot["cholinv_flops"], tot["cholinv_ms"]) / 1e12
        sweep_tf_alone = _rate(clf["cholinv_flops"], clf["cholinv_ms"]) / 1e12
        kb_gbs = _rate(tot["kbuild_bytes"], tot["kbuild_ms"]) / 1e9
        kb_gbs_alone = _rate(clf["kbuild_bytes"], clf["kbuild_ms"]) / 1e9
        pv_tf = _rate(tot["timed_predict_var_flops"], tot["predict_var_ms"]) / 1e12
        kinv_tf = _rate(tot["kinv_flops"], tot["kinv_ms"]) / 1e12
        streamed = tot["kinv_flops"] == 0
        from multifidelity_datafusion_gps_amd._lib import build_id as _build_id
        build_id = _build_id()
        busy, busy_why = mfma_busy(build_id)
        traffic_sweep, traffic_why = pmc_traffic("sweep", args.n, build_id)
        out = {
            "metric": "gp_fit_predict_wall_ms", "value": round(ms_per_step, 2), "unit": "ms",
            "n_gpus": 1 if args.single_device else world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2),
            "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "2-fidelity NARGP (data-driven LF GP + composite-kernel HF GP), d=4, "
                                   "N_lf=N_hf=N*=%d, fp64; %d objective+gradient evaluations per L-BFGS-B run, "
                                   "1 LF run + (1+%d) HF runs, then predict" % (args.n, args.evals, args.restarts),
                       "n": args.n, "evals_per_run": args.evals, "restarts": args.restarts,
                       "evals_issued_rank0_per_step": evals / args.steps,
                       "gpu_ms_per_evaluation": (round(tot["total_ms"] / tot["timed_evals"], 3) if tot["timed_evals"] else None),
                       "wall_ms_per_evaluation": round(ms_per_step * args.steps / max(evals, 1), 3),
                       "restarts_run_as": ("lock-stepped runs over batched evaluations (mfgp_eval_batch): %d lanes, %s live slots"
                                           % (lanes, args.width or "auto")) if lockstep
                                          else "concurrent restarts on %d auxiliary handles (round 3's mode; the default is lock step "
                                               "over batched evaluations)" % max(args.concurrency, 2),
                       "restart_concurrency": None if lockstep else max(args.concurrency, 2), "collectives": collectives,
                       "fit_driver": phases[-1].get("fit_driver"), "jitter_retries_in_timed_steps": retries,
                       "jitter_retries_per_step_rank0": per_step, "jitter_retries_same_in_every_step": retries_stable,
                       "failed_evaluations_in_timed_steps": sum(p["failed_evaluations"] for p in phases),
                       "fitted_noise_variance": phases[-1]["fitted_noise_variance"],
                       "shard_groups_formed": int(getattr(comm, "groups_formed", 0)),   # (communicators created beside the world's: once per group, not per step)
                       "ranks": world, "rccl_ranks": int(engines["lf"].comm_size), "library_build_id": build_id,
                       "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default"),
                       "sharding": sharding_note(world, args.restarts, int(engines["lf"].comm_size)),
                       # first hardware contact decides by measurement: what a collective of the world communicator cost when
                       # it was formed, and what the planner makes of it for the LF level's shared evaluations
                       "sharding_calibration": ({"world_communicator": getattr(comm, "calibration", None),
                                                 "lf_level_decision": engines["lf"].shard_decision()} if world > 1 else None)},
            # the dominant work: ONE sweep per evaluation = Cholesky + triangular inverse%s, timed with HIP events on the
            # engine's main stream around the sweep (the bulk stream joins before the closing event)
            "roofline": {"kernel": "factorisation sweep per evaluation: mfgp_leaf_cholinv_f64 + mfgp_gemm_nt_f64_{t128,t64,chain} on v_mfma_f64_4x4x4_4b "
                                   "(Cholesky N^3/3 + inverse N^3/3%s)" % (" + streamed K^-1 N^3/3" if streamed else ""),
                         "bound": "mfma", "achieved": round(sweep_tf, 2), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(sweep_tf / FP64_PEAK_TFLOPS, 4),
                         "frac_of_bare_mfma_loop": round(sweep_tf / BARE_MFMA_TFLOPS, 4), "bare_mfma_loop": BARE_MFMA_TFLOPS,
                         "mfma_busy": busy, "mfma_busy_source": (MFMA_FILE + " (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES "
                         "SQ_BUSY_CU_CYCLES ... over tools/time_eval.py 8192, committed, taken with this library build %s; not "
                         "measured by this run)" % build_id) if busy is not None else "null: " + busy_why,
                         "traffic": traffic_sweep, "traffic_source": (PMC_FILE + " (rocprofv3 --pmc passes over "
                         "tools/time_eval.py, committed, taken with this library build %s; not measured by this run)" % build_id)
                         if traffic_sweep is not None else "null: " + traffic_why,
                         "launches": int(evals), "flops_per_launch": tot["cholinv_flops"] / max(evals, 1),
                         "flops_are": ("algorithmic flops of one rank's evaluations" if world == 1 else
                                       "rank 0's EXECUTED share: an evaluation shared by a rank group counts what this rank ran (the Cholesky in full or a G-th of it, "
                                       "a G-th of the inverse and of K^-1), not the whole evaluation"),
                         "achieved_is": "all sweep flops of the timed region / its wall time (several evaluations are in flight at once -- "
                                        "%s -- so per-launch intervals overlap)" % ("batched passes on %d lanes" % lanes if lockstep else "%d free-running evaluations" % (1 + max(args.concurrency, 2))),
                         "per_launch_overlapped": {"avg_launch_ms": round(tot["cholinv_ms"] / max(evals, 1), 4),
                                                   "achieved": round(sweep_tf_launch, 2),
                                                   "note": "HIP events around each sweep on its engine's stream; the interval "
                                                           "includes waiting for CUs held by the other evaluations in flight"},
                         "uncontended": {"achieved": round(sweep_tf_alone, 2), "frac": round(sweep_tf_alone / FP64_PEAK_TFLOPS, 4),
                                         "launches": int(clf["evals"]), "avg_launch_ms": round(clf["cholinv_ms"] / max(clf["evals"], 1), 4),
                                         "note": "the LF level's evaluations only: they run alone on the GPU; the HF level's "
                                                 "run batched / beside other lanes"}},
            "roofline_predvar": {"kernel": "mfgp_predvar_f64 (V = K(X*,X) L^-T, one launch per predict)", "bound": "mfma",
                                 "achieved": round(pv_tf, 2), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": round(pv_tf / FP64_PEAK_TFLOPS, 4), "traffic": pmc_traffic("mfgp_predvar_f64", args.n, build_id)[0],
                                 "launches": int(tot["predicts"]), "avg_launch_ms": round(tot["predict_var_ms"] / max(tot["predicts"], 1), 4)},
            "roofline_kbuild": {"kernel": "mfgp_kbuild_rbf2_f64<MODE_TRI> (K(X,X)+noise lower triangle, one launch per evaluation)",
                                "bound": "hbm", "achieved": round(kb_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(kb_gbs / HBM_PEAK_GBS, 4),
                                "traffic": pmc_traffic("mfgp_kbuild_rbf2_f64<0>", args.n, build_id)[0],
                                "launches": int(evals), "avg_launch_ms": round(tot["kbuild_ms"] / max(evals, 1), 4),
                                "uncontended": {"achieved": round(kb_gbs_alone, 1), "frac": round(kb_gbs_alone / HBM_PEAK_GBS, 4),
                                                "avg_launch_ms": round(clf["kbuild_ms"] / max(clf["evals"], 1), 4)}},
            "stage_ms_per_evaluation": {k: round(tot[k] / max(evals, 1), 4)
                                        for k in ("kbuild_ms", "cholinv_ms", "solve_ms", "kinv_ms", "grad_ms")},
            "stage_ms_per_evaluation_uncontended": {k: round(clf[k] / max(clf["evals"], 1), 4)
                                                    for k in ("kbuild_ms", "cholinv_ms", "solve_ms", "kinv_ms", "grad_ms")},
            "result_checksum": {"mean_sum": float(np.sum(mean)), "var_sum": float(np.sum(var))},
        }
        # Amdahl floor of the strong-scaling curve, from THIS run's measurements on rank 0: the LF run and the first HF run ->
        # restart 0 are sequential chains of evaluations no rank count shortens (priced at the uncontended time per evaluation:
        # on N > 1 GPUs rank 0's chain runs alone), the predict is what this run measured for its row share
        alone_ms = (clf["total_ms"] / clf["timed_evals"]) if clf["timed_evals"] else None     # (below N = 4096 nothing is timed)
        lf_ms = float(np.mean([p["lf_ms"] for p in phases]))
        chain_evals = float(np.mean([p["chain_evals"] for p in phases]))
        predict_ms = float(np.mean([p["predict_ms"] for p in phases]))
        out["serial_floor_ms"] = round(lf_ms + chain_evals * alone_ms + predict_ms, 1) if alone_ms is not None else None
        out["serial_floor"] = {"lf_run_ms": round(lf_ms, 1), "hf_chain_evaluations": chain_evals,
                               "ms_per_evaluation_alone": round(alone_ms, 3) if alone_ms is not None else None,
                               "hf_chain_ms": round(chain_evals * alone_ms, 1) if alone_ms is not None else None,
                               "predict_ms_this_run": round(predict_ms, 1), "fit_ms_this_run": round(float(np.mean([p["fit_ms"] for p in phases])), 1),
                               "note": "value cannot fall below serial_floor_ms however many GPUs share the restarts and the "
                                       "predictive rows (the first HF run and restart 0 are one chain of %d + %d evaluations)"
                                       % (args.evals, args.evals)}
        # ... and what the floor becomes where the sequential evaluations are SHARED by a group of GPUs (mfgp_eval_sharded: the
        # LF run by all G ranks, first run -> restart 0 by rank 0 and the ranks that were dealt no restart): one rank's device work
        # of such an evaluation measured HERE (mfgp_dbg_eval_as_rank: rank 0 of G, no exchange), the exchange priced at one xGMI
        # link (153 GB/s) per owner -- a PROJECTION from one GPU, never run over xGMI
        if world == 1 and args.n >= 2048:
            try:
                proj = {}
                e_hf, e_lf = engines["hf"], engines["lf"]
                npad = (args.n + 127) // 128 * 128
                for G in (2, 3, 4, 8):
                    t_hf = sorted(e_hf.dbg_eval_as_rank(np.ones(6), 0.05, 0, G) for _ in range(3))[1]
                    t_lf = sorted(e_lf.dbg_eval_as_rank(np.ones(2), 0.05, 0, G) for _ in range(3))[1]
                    xch = (4.0 * npad * npad / G) / 153e9 * 1e3        # every peer's chunk of the packed rows over its own link (ideal)
                    proj[str(G)] = {"hf_rank_ms": round(t_hf, 3), "lf_rank_ms": round(t_lf, 3), "exchange_ms_model": round(xch, 3)}
                groups = {"2": (2, 1), "4": (4, 1), "8": (8, 3)}       # ranks: (LF group, chain group = rank 0 + ranks without a restart)
                floors = {}
                n_lf = clf["evals"] / args.steps
                for n_gpu, (g_lf, g_ch) in groups.items():
                    lf_eval = proj[str(g_lf)]["lf_rank_ms"] + proj[str(g_lf)]["exchange_ms_model"]
                    ch_eval = alone_ms if g_ch == 1 else proj[str(g_ch)]["hf_rank_ms"] + proj[str(g_ch)]["exchange_ms_model"]
                    floors[n_gpu] = round(n_lf * lf_eval + chain_evals * ch_eval + predict_ms / int(n_gpu), 1)
                out["serial_floor_sharded_projection"] = {
                    "per_rank": proj, "floor_ms_by_gpus": floors,
                    "note": "one GPU's measured share of a sharded evaluation + a bandwidth model of the exchange; the 1 -> 8 GPU "
                            "curve itself has not been measured on hardware"}
            except Exception as ex:  # noqa: BLE001 - diagnostic only
                out["serial_floor_sharded_projection"] = {"error": repr(ex)[:200]}
        if not streamed and 0 < kinv_tf <= FP64_PEAK_TFLOPS:     # (a batched pass has no stamp around its K^-1 launch: nothing to report then)
            out["roofline_kinv"] = {"kernel": "mfgp_kinv_syrk_f64 (stand-alone K^-1 launch)", "bound": "mfma",
                                    "achieved": round(kinv_tf, 2), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": round(kinv_tf / FP64_PEAK_TFLOPS, 4)}
        if phases and "lockstep" in phases[-1]:
            out["lockstep_last_step"] = phases[-1]["lockstep"]
        if power is not None:
            out["power"] = power
        if rowblock is not None:
            out["rowblock_allgather"] = rowblock
        if world == 1 and not args.no_extra:
            # the non-headline configurations of BASELINE.json, driver-run: after the timed region, never part of `value`
            import bench_extra
            out["extra_configs"] = bench_extra.extra_configs()
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args, data, clf["evals"] / args.steps, chf["evals"] / args.steps)
            out["cpu_baseline"] = cb
            out["config"]["gpu_over_cpu"] = round(cb["value"] / ms_per_step, 2)
            if cb.get("lapack_only_ms"):
                # the comparator that is not about numpy temporaries: the same evaluations priced at the measured dpotrf + dtrtri + dpotri
                out["config"]["gpu_over_cpu_lapack_only"] = round(cb["lapack_only_ms"] / ms_per_step, 2)
        print(json.dumps(out), flush=True)
    comm.barrier()
    if world > 1 and comm.transport == "rccl":
        for e_ in engines.values():       # every rank still alive: destroy the communicators (world, chain group) before anyone exits
            if e_.comm_size > 1:
                e_.comm_destroy()
        comm.barrier()
    comm.close()


if __name__ == "__main__":
    try:
        main()
    except ConnectionError as ex:
        if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
            raise
        # a peer vanished mid-collective: say so with a code of its own, so that the launcher reports the rank that failed first
        sys.stderr.write("bench.py: rank %s lost a peer: %s\n" % (os.environ.get("RANK", "?"), ex))
        sys.stderr.flush()
        os._exit(EXIT_PEER_LOST)
