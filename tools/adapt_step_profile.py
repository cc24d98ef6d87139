"""Where one acquisition step of BASELINE cfg5 at its last size goes (NARGP d = 4, N_lf = 16384, N_hf = 8128; the batched DIRECT of
adaptation_maximizers/direct.py): every Engine.predict / predict_chained / append_row call of ONE step with its row count and host-clock
time, as a histogram by row count.

    python tools/adapt_step_profile.py [steps]"""
import os, sys, time
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multifidelity_datafusion_gps_amd as mf
from multifidelity_datafusion_gps_amd import _lib
import bench_extra as bx

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
log = []
def wrap(name):
    orig = getattr(_lib.Engine, name)
    def f(self, *a, **k):
        t0 = time.perf_counter()
        r = orig(self, *a, **k)
        X = a[1] if name == "predict_chained" else a[0]
        log.append((name, int(np.atleast_2d(X).shape[0]) if name != "append_row" else 1, (time.perf_counter() - t0) * 1e3))
        return r
    setattr(_lib.Engine, name, f)
for n in ("predict", "predict_chained", "append_row"):
    wrap(n)

class BudgetNARGP(mf.NARGP):
    lf_max_iters = first_run_max_iters = restart_max_iters = 2
    eval_cap = 2
rng = np.random.default_rng(2)
X_lf = rng.uniform(size=(16384, 4))
m = BudgetNARGP(4, bx._hf_4d, None, lf_X=X_lf, lf_Y=bx._lf_4d(X_lf), lf_hf_adapt_ratio=0, seed=2)
m.num_restarts = 1; m.eps = 0.0
m.fit(rng.uniform(size=(8192 - 64, 4)))
m.adapt_maximizer = mf.DIRECT1Maximizer()
m.adapt(1, reoptimize=False)
log.clear()
t0 = time.perf_counter()
m.adapt(steps, reoptimize=False)
dt = (time.perf_counter() - t0) * 1e3
by = defaultdict(lambda: [0, 0.0])
for name, rows, ms in log:
    b = by[(name, rows if rows <= 4 else (8 if rows <= 8 else (16 if rows <= 16 else (32 if rows <= 32 else (64 if rows <= 64 else (128 if rows <= 128 else 1 << 20))))))]
    b[0] += 1; b[1] += ms
print("%d steps: %.2f ms per step, %d engine calls per step, %.2f ms of the step inside them" % (steps, dt / steps, len(log) / steps, sum(l[2] for l in log) / steps))
for k in sorted(by):
    print("  %-16s rows <= %-7d %4d calls per step  %.3f ms each  %.2f ms per step" % (k[0], k[1], by[k][0] / steps, by[k][1] / by[k][0], by[k][1] / steps))
m.close()
