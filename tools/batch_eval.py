"""One batched objective+gradient pass (mfgp_eval_batch) against B single evaluations: ms per pass and per evaluation, executed
TFLOP/s (Np^3 flops per evaluation).  usage: batch_eval.py [N ...]   (env BATCHES="1 2 3 4 6 8")"""
import os
os.environ.setdefault("MFGP_HW_QUEUES", "2")   # opt-in since round 4 (2 hardware queues per priority: profiles/r03_hw_queues.txt)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from multifidelity_datafusion_gps_amd._lib import Engine  # noqa: E402
from tests import cases  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [512, 1024, 2048, 4096, 8192]
batches = [int(b) for b in os.environ.get("BATCHES", "1 2 3 4 6 8").split()]
rng = np.random.default_rng(0)
parts = cases.composite(4, 1)
print("# tools/batch_eval.py: ms per batched pass [ms per evaluation, TFLOP/s executed on Np^3 flops per evaluation]")
for N in sizes:
    X = rng.uniform(size=(N, 4))
    Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
    Y = cases.hf_4d(X)
    e = Engine(0)
    e.set_data(Xa, Y)
    e.set_kernel(parts)
    theta, noise = np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.01 * Y.var()
    reps = 3 if N >= 8192 else (6 if N >= 4096 else 20)
    Np = (N + 127) // 128 * 128
    e.eval(theta, noise)
    t0 = time.perf_counter()
    for _ in range(reps):
        e.eval(theta, noise)
    single = (time.perf_counter() - t0) / reps * 1e3
    line = "N=%d  single eval %.3f ms (%.1f TF)" % (N, single, Np ** 3 / single / 1e9)
    for B in batches:
        thetas = np.tile(theta, (B, 1)) * np.linspace(0.9, 1.1, B)[:, None]
        e.eval_batch(thetas, np.full(B, noise))
        t0 = time.perf_counter()
        for _ in range(reps):
            e.eval_batch(thetas, np.full(B, noise))
        ms = (time.perf_counter() - t0) / reps * 1e3
        line += " | B=%d %.3f [%.3f, %.1f TF]" % (B, ms, ms / B, B * Np ** 3 / ms / 1e9)
    print(line, flush=True)
    e.close()
