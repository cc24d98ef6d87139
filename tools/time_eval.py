"""quick per-stage timing of one objective+gradient evaluation and one predict (GPU box)"""
import sys, os
import os as _os; _os.environ.setdefault("MFGP_TIMING", "1")   # start / end stamps of a call at every size (off by default below Np = 4096)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multifidelity_datafusion_gps_amd._lib import Engine
from tests import cases

def main():
    ns = [int(a) for a in sys.argv[1:]] or [8192]
    e = Engine(0)
    for N in ns:
        rng = np.random.default_rng(N)
        X = rng.uniform(size=(N, 4)); Y = cases.hf_4d(X)
        Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
        parts, theta, noise = cases.composite(4, 1), np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.01 * Y.var()
        e.set_data(Xa, Y); e.set_kernel(parts)
        for _ in range(3):   # warm-up: first use plans and allocates
            e.eval(theta, noise)
        acc = {}
        reps = 5
        for _ in range(reps):
            e.eval(theta, noise)
            t = e.timings()
            for k, v in t.items():
                acc[k] = acc.get(k, 0) + v / reps
        m, v = e.predict(Xa[: min(N, 8192)])
        tp = e.timings()
        Np = (N + 127) // 128 * 128
        for k in ("kbuild_ms", "cholinv_ms", "kinv_ms"):   # per-stage stamps are off below Np = 4096 (MFGP_STAGE_TIMING=1 forces them)
            acc[k] = max(acc[k], 1e-9)
        print("N=%d  total %.3f ms | kbuild %.3f (%.0f GB/s) cholinv %.3f (%.1f TF) solve %.3f kinv %.3f (%.1f TF) grad %.3f | launches %d | predict panel %.3f var %.3f (%.1f TF)" % (
            N, acc["total_ms"], acc["kbuild_ms"], acc["kbuild_bytes"] / acc["kbuild_ms"] / 1e6, acc["cholinv_ms"],
            acc["cholinv_flops"] / acc["cholinv_ms"] / 1e9, acc["solve_ms"], acc["kinv_ms"], acc["kinv_flops"] / acc["kinv_ms"] / 1e9,
            acc["grad_ms"], acc["n_launches"], tp["predict_panel_ms"], tp["predict_var_ms"],
            float(Np) * Np * min(N, 8192) / max(tp["predict_var_ms"], 1e-9) / 1e9))

main()
