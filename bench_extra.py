#!/usr/bin/env python3
"""bench_extra.py -- the BASELINE.json configurations that are NOT the headline (cfg2, cfg3, cfg5), measured in the same driver
run as bench.py's line, OUTSIDE its timed region and outside `value` (every such figure used to be builder-run).

    cfg2  single-fidelity GP, d = 3, N = 4096, RBF: factorisation (K build + Cholesky + inverse) and one
          objective+gradient evaluation, alone and inside a batch of 4 (mfgp_eval_batch), each with its fraction of the fp64 peak
    cfg3  2-fidelity NARGP, d = 4, N_lf = 16384 / N_hf = 4096: the LF run and the HF recipe (1 + 6 runs x 20 evaluations), predict
    cfg5  one adaptation step at N_hf = 8192 (N_lf = 16384) in three forms -- the reference's N* = 1 callbacks through Gablonsky's
          DIRECT (src/adaptation_maximizers/DIRECT1_maximizer.py:18-27), the batched DIRECT, a 65536-row candidate panel -- each
          followed by a rank-1 append; beside them the two bandwidth-bound kernels of the loop on their own: one N* = 1 predict
          call (variance stage = one read of the 4 Np (Np + 1)-byte triangle of L^-1) and one append (two reads)

`python bench_extra.py` prints the same dictionary on its own (profiles/r06_extra_configs.json is that output; the rocprofv3
kernel statistics of the same command are profiles/r06_extra_configs_kernel_stats.csv).  About 10 s of GPU time."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6
HBM_PEAK_GBS = 8000.0


def _hf_3d(x):
    return np.prod(np.sin(np.pi * x[:, :3]), axis=1) + 5.0


def _hf_4d(x):
    return (np.prod(np.sin(np.pi * x[:, :4]), axis=1) + 5.0)[:, None]


def _lf_4d(x):
    return _hf_4d(x) - 0.25 * (np.sin(x[:, 0] * np.pi * 0.1) + np.sin(x[:, 1] * np.pi * 0.05)
                               + np.sin(x[:, 2] * 0.15 * np.pi) + np.sin(x[:, 3] * 0.2 * np.pi))[:, None]


def _best(fn, reps):
    best = np.inf
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def cfg2(Engine):
    rng = np.random.default_rng(1)
    N = 4096
    X = rng.uniform(size=(N, 3)); Y = _hf_3d(X)
    th, nz = np.array([1.0, 0.3]), 0.01 * Y.var()
    e = Engine(0)
    try:
        e.set_data(X, Y); e.set_kernel([(0, 0, 3, 0)])
        e.eval(th, nz); e.factorize(th, nz)                                   # plans, first launches
        fac = _best(lambda: e.factorize(th, nz), 5)
        tf = e.timings()
        ev = _best(lambda: e.eval(th, nz), 5)
        te = e.timings()
        ths = np.array([th * s for s in (1.0, 1.01, 0.99, 1.02)])
        e.eval_batch(ths, nz)
        bt = _best(lambda: e.eval_batch(ths, nz), 5)
        fl_fac, fl_ev = 2.0 * N ** 3 / 3, float(N) ** 3
        return {"workload": "single GP, d=3, N=4096, RBF, theta=(1, 0.3), noise=0.01 Var(y)",
                "factorisation_call_ms": round(fac, 4), "factorisation_sweep_ms": round(tf["cholinv_ms"], 4), "kbuild_ms": round(tf["kbuild_ms"], 4),
                "factorisation_flops": fl_fac, "factorisation_frac": round(fl_fac / (fac * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4),
                "objective_gradient_call_ms": round(ev, 4), "objective_gradient_device_ms": round(te["total_ms"], 4),
                "objective_gradient_flops": fl_ev, "objective_gradient_frac": round(fl_ev / (ev * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4),
                "batch_of_4_pass_ms": round(bt, 4), "batch_of_4_ms_per_evaluation": round(bt / 4, 4),
                "batch_of_4_frac": round(4 * fl_ev / (bt * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4),
                "timed_as": "host clock around the C-ABI call (synchronised), best of 5 after a warm-up"}
    finally:
        e.close()


def cfg3_cfg5(mf, evals=20):
    class BudgetNARGP(mf.NARGP):
        lf_max_iters = first_run_max_iters = restart_max_iters = evals
        eval_cap = evals

    out = {}
    rng = np.random.default_rng(2)
    n_lf, n_hf = 16384, 4096
    X_lf = rng.uniform(size=(n_lf, 4)); X_hf = rng.uniform(size=(n_hf, 4)); Xs = rng.uniform(size=(4096, 4))
    t0 = time.perf_counter()
    # (lf_hf_adapt_ratio = 0: the acquisitions below grow the high-fidelity set only -- the low-fidelity level keeps its 16384 rows -- while
    # predictions stay on the data-driven path with the level hand-over on the device)
    m = BudgetNARGP(4, _hf_4d, None, lf_X=X_lf, lf_Y=_lf_4d(X_lf), lf_hf_adapt_ratio=0, seed=2)
    t1 = time.perf_counter()
    m.fit(X_hf)                                                               # first fit on these handles: plans, slabs
    t2 = time.perf_counter()
    m.fit(X_hf)
    t3 = time.perf_counter()
    n_ev = m.hf_model.n_evals
    mean, var = m.predict(Xs)
    t4 = time.perf_counter()
    mean, var = m.predict(Xs)
    t5 = time.perf_counter()
    out["cfg3"] = {"workload": "2-fidelity NARGP, d=4, N_lf=16384 (data-driven LF GP, one run) / N_hf=4096 (1 + 6 runs), %d evaluations per run" % evals,
                   "lf_fit_ms": round((t1 - t0) * 1e3, 1), "lf_evaluations": int(m.lf_model.n_evals),
                   "hf_fit_first_ms": round((t2 - t1) * 1e3, 1), "hf_fit_ms": round((t3 - t2) * 1e3, 1), "hf_evaluations": int(n_ev),
                   "hf_fit_frac": round(n_ev * float(n_hf) ** 3 / (t3 - t2) / 1e12 / FP64_PEAK_TFLOPS, 4),
                   "predict_4096_ms": round((t5 - t4) * 1e3, 2),
                   "mse": float(np.mean((mean - _hf_4d(Xs)) ** 2)),
                   "hf_fit_ms_is": "the second of two identical fits on the same handles (the first also builds the plans and the batch slab)"}
    # ---- cfg5 at its last size: N_hf = 8192 - 64 (64 padding slots: the appends below never cross a 128-row boundary) -------------
    n8 = 8192 - 64
    m.eval_cap = m.lf_max_iters = m.first_run_max_iters = m.restart_max_iters = 2     # a token fit: the acquisition is what is timed
    m.num_restarts = 1
    m.eps = 0.0                                                                         # never stop early
    m.fit(rng.uniform(size=(n8, 4)))
    Np = 8192
    tri = 4.0 * Np * (Np + 1)
    c5 = {"workload": "NARGP d=4, N_lf=16384, N_hf=%d (Np=8192), one acquisition + rank-1 append per step" % n8}
    for name, mx, steps in (("callbacks_nstar1_gablonsky_direct1", mf.DIRECT1Maximizer(faithful=True), 1),
                            ("batched_direct1", mf.DIRECT1Maximizer(), 2),
                            ("panel_65536", mf.PanelMaximizer(n_candidates=65536, seed=1), 2)):
        m.adapt_maximizer = mx
        m.adapt(1, reoptimize=False)                                                   # warm-up (panel draw, buffers)
        t0 = time.perf_counter()
        m.adapt(steps, reoptimize=False)
        dt = (time.perf_counter() - t0) * 1e3 / steps
        info = getattr(mx, "last_info", None) or {}
        c5[name + "_ms_per_step"] = round(dt, 2)
        if "nf" in info:
            c5[name + "_acquisition_evaluations"] = int(info["nf"])
    c5["panel_variance_product_frac"] = round(float(Np) * Np * 65536 / (c5["panel_65536_ms_per_step"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4)
    # the loop's two bandwidth-bound kernels on their own, on the high-fidelity handle
    e = m.hf_model._engine
    Xa = m.hf_model.X
    x1 = Xa[:1] + 0.01
    for _ in range(5):
        e.predict(x1)
    reps, var_ms = 200, 0.0
    t0 = time.perf_counter()
    for _ in range(reps):
        e.predict(x1)
    call = (time.perf_counter() - t0) * 1e3 / reps       # the call alone; the stage stamps are read in a loop of their own
    for _ in range(50):
        e.predict(x1)
        var_ms += e.timings()["predict_var_ms"]
    var_ms /= 50
    c5["nstar1_predict_call_ms"] = round(call, 4)
    c5["nstar1_variance_stage_ms"] = round(var_ms, 4)
    c5["nstar1_variance_stage_GBps"] = round(tri / (var_ms * 1e-3) / 1e9, 1) if var_ms > 0 else None
    c5["nstar1_variance_stage_frac"] = round(tri / (var_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if var_ms > 0 else None
    c5["variance_stage_is"] = ("mfgp_predv_rows_f64 (V = k(x*,X) L^-T on the VALU behind one coalesced read of the triangle, + the mean) "
                               "+ mfgp_predv_finish_f64, between two HIP events on the engine's stream; algorithmic bytes = 4 Np (Np + 1)")
    # the call the reference's loop issues per acquisition evaluation (src/abstractMFGP.py:317-359 -> predict of the NARGP model,
    # src/MFDataFusion.py:106-156): the low-fidelity mean of the test row (N_lf = 16384 columns) and the high-fidelity predict, the
    # hand-over on the device
    xb = rng.uniform(size=(1, 4))
    for _ in range(5):
        m.predict(xb)
    t0 = time.perf_counter()
    for _ in range(reps):
        m.predict(xb)
    c5["nstar1_model_predict_call_ms"] = round((time.perf_counter() - t0) * 1e3 / reps, 4)
    c5["nstar1_model_predict_is"] = ("NARGP.predict of one row: mfgp_predict_chained (LF panel row + mean over 16384 columns, augmented row, HF panel, "
                                     "variance product, finish: 7 launches + one upload), host clock around the Python call")
    ts = []
    Xn = rng.uniform(size=(12, 4))
    Xn = np.hstack([Xn, m.lf_model.predict(Xn)[0]]) if Xa.shape[1] == 5 else Xn
    Yn = _hf_4d(Xn)[:, 0]
    for i in range(12):
        t0 = time.perf_counter()
        ok = e.append_row(Xn[i], Yn[i])
        ts.append((time.perf_counter() - t0) * 1e3)
        if not ok:
            break
    ts = ts[2:]
    if ts:
        c5["append_call_ms"] = round(float(np.median(ts)), 4)
        c5["append_GBps"] = round(2 * tri / (float(np.median(ts)) * 1e-3) / 1e9, 1)
        c5["append_frac"] = round(2 * tri / (float(np.median(ts)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        c5["append_is"] = "mfgp_append_row: panel row, l = X k, w = X^T l (two reads of the triangle), finish incl. alpha; host clock around the call, median of %d" % len(ts)
    out["cfg5"] = c5
    m.close()
    return out


def extra_configs():
    """-> {"cfg2": ..., "cfg3": ..., "cfg5": ..., "seconds": ...}; never raises (an error is reported under its key)"""
    t0 = time.perf_counter()
    out = {}
    stamps = os.environ.get("MFGP_TIMING")
    os.environ["MFGP_TIMING"] = "1"      # (read when a handle takes its data: a predict of <= 64 rows is stamped only on request)
    try:
        from multifidelity_datafusion_gps_amd._lib import Engine, build_id
        import multifidelity_datafusion_gps_amd as mf
        out["library_build_id"] = build_id()
        out["cfg2"] = cfg2(Engine)
        out.update(cfg3_cfg5(mf))
    except Exception as ex:  # noqa: BLE001 - diagnostic block: never fails the bench line
        import traceback
        out["error"] = repr(ex)[:300]
        out["traceback"] = traceback.format_exc()[-600:]
    finally:
        if stamps is None:
            os.environ.pop("MFGP_TIMING", None)
        else:
            os.environ["MFGP_TIMING"] = stamps
    out["seconds"] = round(time.perf_counter() - t0, 1)
    out["note"] = "measured after the timed region of bench.py, outside `value`; peaks: %.1f TFLOP/s fp64, %.0f GB/s HBM" % (FP64_PEAK_TFLOPS, HBM_PEAK_GBS)
    return out


if __name__ == "__main__":
    os.environ.setdefault("MFGP_HW_QUEUES", "2")
    print(json.dumps(extra_configs()))
