"""Latency of the adaptation loop's own calls -- one predict call (mean + variance) with N* = 1 .. 64
test rows, the DIRECT callback shape of the reference (src/adaptation_maximizers/scipydirect_wrapper.py:22-24), and one rank-1
append -- with the GB/s of the triangle read behind each: the variance product reads the 4 Np (Np + 1)-byte lower part of the
mirrored inverse once, an append reads it twice (l = X k, w = X^T l).

    python tools/adapt_latency.py [N ...]            (default 2048 4096 8192; N is rounded down by 64 so that appends fit)
    ADAPT_REPS=200                                    calls per figure

Writes one JSON object per line (for profiles/), after the human-readable lines."""
import json
import os
import sys
import time

os.environ.setdefault("MFGP_TIMING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multifidelity_datafusion_gps_amd._lib import Engine, build_id
from tests import cases

REPS = int(os.environ.get("ADAPT_REPS", "200"))
sizes = [int(a) for a in sys.argv[1:]] or [2048, 4096, 8192]
e = Engine(0)
out = []
for N in sizes:
    n0 = N - 64                                    # 64 padding slots: the appends below fit without a re-upload
    rng = np.random.default_rng(N)
    X = rng.uniform(size=(N, 4)); Y = cases.hf_4d(X)
    Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
    e.set_data(Xa[:n0], Y[:n0]); e.set_kernel(cases.composite(4, 1))
    theta = np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8])
    e.factorize(theta, 0.01)
    Np = (n0 + 127) // 128 * 128
    tri = 4.0 * Np * (Np + 1)                      # bytes of the lower part incl. the diagonal
    rec = {"N": n0, "Np": Np, "triangle_MB": tri / 1e6, "library_build_id": build_id()}
    for ns in (1, 2, 4, 8, 16, 17, 32, 64):
        Xs = Xa[:ns] + 0.01
        for _ in range(5):
            e.predict(Xs)
        t0 = time.perf_counter()
        pan = var = 0.0
        for _ in range(REPS):
            e.predict(Xs)
            t = e.timings()
            pan += t["predict_panel_ms"]; var += t["predict_var_ms"]
        dt = (time.perf_counter() - t0) / REPS * 1e3
        pan /= REPS; var /= REPS
        print("N=%5d N*=%3d  %.4f ms per predict call (panel %.4f, variance stage %.4f ms = %.2f TB/s of triangle)"
              % (n0, ns, dt, pan, var, tri / (var * 1e-3) / 1e12 if var > 0 else 0.0), flush=True)
        rec["predict_call_ms_nstar%d" % ns] = round(dt, 5)
        rec["variance_stage_ms_nstar%d" % ns] = round(var, 5)
        rec["variance_stage_TBps_nstar%d" % ns] = round(tri / (var * 1e-3) / 1e12, 3) if var > 0 else None      # (MFGP_TIMING=0: no stamps)
    # appends: 48 of the 64 slots, each timed by the host clock around the call (the call ends synchronised)
    ts = []
    for i in range(48):
        t0 = time.perf_counter()
        ok = e.append_row(Xa[n0 + i], Y[n0 + i])
        ts.append(time.perf_counter() - t0)
        assert ok
    ts = np.array(ts[4:]) * 1e3
    print("N=%5d append  %.4f ms median (min %.4f, max %.4f) = %.2f TB/s over two passes of the triangle"
          % (n0, np.median(ts), ts.min(), ts.max(), 2 * tri / (np.median(ts) * 1e-3) / 1e12), flush=True)
    rec["append_ms_median"] = round(float(np.median(ts)), 5)
    rec["append_ms_min"] = round(float(ts.min()), 5)
    rec["append_two_pass_TBps"] = round(2 * tri / (float(np.median(ts)) * 1e-3) / 1e12, 3)
    # the appended state against a fresh factorisation of the same rows
    f_app = e.nlml(); a_app = e.get_alpha(); m_app, v_app = e.predict(Xa[:7] + 0.02)
    e.set_data(Xa[:n0 + 48], Y[:n0 + 48]); e.factorize(theta, 0.01)
    f_new = e.nlml(); a_new = e.get_alpha(); m_new, v_new = e.predict(Xa[:7] + 0.02)
    rec["append_vs_fresh"] = {"nlml_rel": abs(f_app - f_new) / abs(f_new), "alpha_rel": float(np.abs(a_app - a_new).max() / np.abs(a_new).max()),
                              "mean_abs": float(np.abs(m_app - m_new).max()), "var_abs": float(np.abs(v_app - v_new).max())}
    print("   48 appends vs fresh factorisation:", rec["append_vs_fresh"], flush=True)
    out.append(rec)
for rec in out:
    print(json.dumps(rec))
