"""What one GPU of a G-GPU group spends on ONE sharded evaluation (mfgp_dbg_eval_as_rank: rank r's device work without the
exchange steps) -- the input of DESIGN.md section 7's projection of the multi-GPU serial floor.  MEASURED ON ONE GPU: the
exchange (4 Np^2 bytes of packed rows per evaluation in one all-gather; P + 1 sums per tile in one all-reduce) is priced
separately from link bandwidth, it has never run over xGMI.   usage: shard_projection.py [N ...]
With MFGP_DIST_CHOL=1 in the environment the ranks' plans are those of the DISTRIBUTED Cholesky (plan.h Shard::dist: every rank only its own
rows of the panels and trailing updates; without a communicator the exchange steps move nothing, so the numbers a rank computes are garbage
and only its device TIME means something): the per-rank compute a 1-D block-cyclic factorisation leaves, to be held against its two
collectives per block column."""
import os
os.environ.setdefault("MFGP_HW_QUEUES", "2")
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from multifidelity_datafusion_gps_amd._lib import Engine  # noqa: E402
from tests import cases  # noqa: E402

print("# tools/shard_projection.py: ms of device work per rank for one objective+gradient evaluation, no exchange (median of 5)")
for N in [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384]:
    rng = np.random.default_rng(N)
    X = rng.uniform(size=(N, 4))
    Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
    Y = cases.hf_4d(X)
    e = Engine(0)
    e.set_data(Xa, Y)
    e.set_kernel(cases.composite(4, 1))
    theta, noise = np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.01 * Y.var()
    for _ in range(2):
        e.eval(theta, noise)
    t = []
    for _ in range(5):
        e.eval(theta, noise)
        t.append(e.timings()["total_ms"])
    line = "N=%d  mfgp_eval %.2f ms" % (N, sorted(t)[2])
    Np = (N + 127) // 128 * 128
    for G in (1, 2, 3, 4, 8):
        per = []
        for r in sorted({0, G // 2, G - 1}):
            e.dbg_eval_as_rank(theta, noise, r, G)
            ms = sorted(e.dbg_eval_as_rank(theta, noise, r, G) for _ in range(5))[2]
            per.append(ms)
        # exchange: the packed upper part of S, 4 Np^2 bytes in all; every rank receives the other ranks' chunks, each over that
        # owner's own xGMI link (153 GB/s) if the all-gather uses all links at once -- the IDEAL figure
        xch = 0.0 if G == 1 else (4.0 * Np * Np / G) / 153e9 * 1e3
        if os.environ.get("MFGP_DIST_CHOL") == "1" and G > 1:
            nb = Np // 128
            vol = 8.0 * 128 * sum(2 * 128 + (Np - 128 * (c + 1)) for c in range(nb))      # bytes of the Cholesky's own exchange steps, in all
            line += " | G=%d dist: %.2f ms (ranks %s) + %d collectives on the chain, %.0f MB in all (+ the rows of X^T ~%.2f ms)" % (
                G, max(per), "/".join("%.2f" % p for p in per), 2 * nb - 1, vol / 1e6, xch)
        else:
            line += " | G=%d: %.2f ms (ranks %s) + exchange ~%.2f ms" % (G, max(per), "/".join("%.2f" % p for p in per), xch)
    print(line, flush=True)
    e.close()
