"""Which engine handles of a process share a hardware queue?  E engines are created back to back (as bench.py creates lf, hf,
hf#1, hf#2); every PAIR runs evaluations at a chain-bound size concurrently: a pair on separate queues overlaps (~0.5 ms per
evaluation at N = 2048), a pair on one queue serialises (~0.87).  usage: queue_pairs.py [E=4] [N=2048]"""
import itertools
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from multifidelity_datafusion_gps_amd._lib import Engine  # noqa: E402
from tests import cases  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
first_use = os.environ.get("FIRST_USE", "creation")     # "reverse": touch the engines in reverse order first
engs = [Engine(0) for _ in range(E)]
rng = np.random.default_rng(N)
X = rng.uniform(size=(N, 4)); Y = cases.hf_4d(X)
Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
parts, theta, noise = cases.composite(4, 1), np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.01 * Y.var()
order = list(reversed(range(E))) if first_use == "reverse" else list(range(E))
for k in order:
    engs[k].set_data(Xa, Y); engs[k].set_kernel(parts); engs[k].eval(theta, noise); engs[k].eval(theta, noise)


def run(ids, reps=20):
    def work(e):
        for _ in range(reps):
            e.eval(theta, noise)
    ts = [threading.Thread(target=work, args=(engs[k],)) for k in ids]
    t0 = time.perf_counter()
    [t.start() for t in ts]; [t.join() for t in ts]
    return (time.perf_counter() - t0) * 1e3 / (reps * len(ids))


print("GPU_MAX_HW_QUEUES=%s first use: %s; alone: %.3f ms" % (os.environ.get("GPU_MAX_HW_QUEUES"), first_use, run([0])))
for a, b in itertools.combinations(range(E), 2):
    print("  engines %d + %d: %.3f ms per evaluation" % (a, b, run([a, b])))
