"""Aggregate throughput of K objective+gradient evaluations in flight (K engine handles, one host thread each), per size:
wall ms per evaluation alone and with K in flight.  usage: inflight_throughput.py [--reps R] [--k 1,2,3,4] N ..."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from multifidelity_datafusion_gps_amd._lib import Engine  # noqa: E402
from tests import cases  # noqa: E402


def main():
    argv = sys.argv[1:]
    reps, ks = 12, [1, 2, 3, 4]
    while argv and argv[0].startswith("--"):
        if argv[0] == "--reps":
            reps = int(argv[1])
        elif argv[0] == "--k":
            ks = [int(v) for v in argv[1].split(",")]
        argv = argv[2:]
    sizes = [int(a) for a in argv] or [4096]
    engs = [Engine(0) for _ in range(max(ks))]
    for N in sizes:
        rng = np.random.default_rng(N)
        X = rng.uniform(size=(N, 4)); Y = cases.hf_4d(X)
        Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
        parts, theta, noise = cases.composite(4, 1), np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.01 * Y.var()
        for e in engs:
            e.set_data(Xa, Y); e.set_kernel(parts)
            e.eval(theta, noise); e.eval(theta, noise)
        line = "N=%d" % N
        for K in ks:
            def work(e):
                for _ in range(reps):
                    e.eval(theta, noise)
            ts = [threading.Thread(target=work, args=(engs[k],)) for k in range(K)]
            t0 = time.perf_counter()
            [t.start() for t in ts]; [t.join() for t in ts]
            dt = time.perf_counter() - t0
            line += "  K=%d: %.3f ms/eval (each %.3f)" % (K, dt * 1e3 / (K * reps), dt * 1e3 / reps)
        print(line, flush=True)


main()
