"""lanes x batch: L engine handles driven by L host threads, each running batched passes of B evaluations back to back --
aggregate ms per evaluation.  usage: lanes_batch.py N "L:B L:B ..." """
import os
os.environ.setdefault("MFGP_HW_QUEUES", "2")   # opt-in since round 4 (2 hardware queues per priority: profiles/r03_hw_queues.txt)
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from multifidelity_datafusion_gps_amd._lib import Engine  # noqa: E402
from tests import cases  # noqa: E402

N = int(sys.argv[1])
combos = [tuple(int(v) for v in c.split(":")) for c in sys.argv[2].split()]
rng = np.random.default_rng(0)
X = rng.uniform(size=(N, 4))
Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
Y = cases.hf_4d(X)
theta, noise = np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.01 * Y.var()
maxl = max(l for l, _ in combos)
engines = []
for _ in range(maxl):
    e = Engine(0); e.set_data(Xa, Y); e.set_kernel(cases.composite(4, 1)); engines.append(e)
Np = (N + 127) // 128 * 128
reps = 4 if N >= 8192 else (10 if N >= 4096 else 30)
line = "N=%d:" % N
for L, B in combos:
    thetas = np.tile(theta, (B, 1)) * np.linspace(0.9, 1.1, B)[:, None]
    nz = np.full(B, noise)
    for e in engines[:L]:
        e.eval_batch(thetas, nz)
    bar = threading.Barrier(L + 1)

    def work(e):
        bar.wait()
        for _ in range(reps):
            e.eval_batch(thetas, nz)
        bar.wait()
    ts = [threading.Thread(target=work, args=(e,)) for e in engines[:L]]
    for t in ts:
        t.start()
    bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
    for t in ts:
        t.join()
    per = dt / (reps * L * B) * 1e3
    line += "  %dx%d: %.3f ms/eval (%.1f TF)" % (L, B, per, Np ** 3 / per / 1e9)
print(line, flush=True)
