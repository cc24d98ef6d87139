"""A mid-size fit as a user of the reference would run it (the ARD recipe: 1 + 6 L-BFGS-B runs, here with a budget of --evals
evaluations per run): sequential restarts (the reference's order), concurrent ones on auxiliary handles (round 3) and the
LOCK-STEPPED runs over one batched evaluation per round (round 4, the default).
usage: midsize_fit.py [--evals E] N_hf [N_hf ...]"""
import os
os.environ.setdefault("MFGP_HW_QUEUES", "2")   # opt-in since round 4 (2 hardware queues per priority: profiles/r03_hw_queues.txt)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import multifidelity_datafusion_gps_amd as mf  # noqa: E402
from tests import cases  # noqa: E402


def col(f):
    return lambda x: f(x)[:, None]


argv = sys.argv[1:]
evals = 50
if argv and argv[0] == "--evals":
    evals = int(argv[1]); argv = argv[2:]
MODES = [("sequential", dict(restart_lockstep=False, restart_concurrency=1)),
         ("concurrent 2", dict(restart_lockstep=False, restart_concurrency=2)),
         ("lockstep 1 lane", dict(restart_lockstep=True, lockstep_lanes=1)),
         ("lockstep 2 lanes", dict(restart_lockstep=True, lockstep_lanes=2)),
         ("lockstep, lanes by size (default)", dict(restart_lockstep=True)),
         ("2 lanes, a thread per run", dict(restart_lockstep=True, lockstep_lanes=2, lockstep_threads=True)),

         ("2 lanes w4", dict(restart_lockstep=True, lockstep_lanes=2, lockstep_width=4)),
         ("3 lanes", dict(restart_lockstep=True, lockstep_lanes=3))]
print("# python3 tools/midsize_fit.py --evals %d ...: one fit of the reference's ARD recipe (1 + 6 L-BFGS-B runs, %d evaluations each), HF level only timed" % (evals, evals))
for n_hf in [int(a) for a in argv] or [1024]:
    line = "N_hf=%d (N_lf=%d):" % (n_hf, 2 * n_hf)
    ref = None
    for name, kw in MODES:
        M = type("M", (mf.NARGP,), dict(lf_max_iters=evals, first_run_max_iters=evals, restart_max_iters=evals, eval_cap=evals, **kw))
        rng = np.random.default_rng(1)
        X_lf = rng.uniform(size=(2 * n_hf, 4))
        m = M(4, col(cases.hf_4d), None, lf_X=X_lf, lf_Y=col(cases.lf_4d)(X_lf), seed=3)
        X = rng.uniform(size=(n_hf, 4))
        m.fit(X)                      # warm: plans, allocations
        dt = np.inf
        for _ in range(3 if n_hf <= 1024 else 1):      # (small fits: best of three -- a 15 ms fit feels every scheduling hiccup of the host)
            t0 = time.perf_counter()
            m.fit(X)
            dt = min(dt, time.perf_counter() - t0)
        th = np.array([p.value for p in m.hf_model.parameters()])
        if ref is None:
            ref = th
        lanes = getattr(m, "last_lockstep_lanes", None) if kw.get("restart_lockstep") else None
        eng = (" [in eval_batch: %s ms, rounds %s]" % ("/".join("%.0f" % (l.engine_s * 1e3) for l in lanes), "/".join(str(l.rounds) for l in lanes))
               if lanes and os.environ.get("MIDSIZE_DETAIL") else "")
        line += "  %s: %.0f ms (%d evals)%s%s" % (name, dt * 1e3, m.hf_model.n_evals, eng, "" if np.array_equal(th, ref) else " DIFFERENT theta")
        m.close()
    print(line, flush=True)
print("# identical fitted hyper-parameters in every column (bit for bit) unless a column says otherwise")
