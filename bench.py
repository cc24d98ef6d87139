#!/usr/bin/env python3
"""bench.py -- the north-star measurement: fit + predict wall-time of a 2-fidelity NARGP at
N_lf = N_hf = N* = 8192, d = 4, fp64 (BASELINE.json `metric`; SURVEY.md 8(d) "North-star").

One STEP = one complete pass of the hot path over one batch of synthetic input:
  level 1 (low fidelity):  GPRegression(X_lf, Y_lf), RBF(4)           -> optimize()            1 L-BFGS-B run
  level 2 (high fidelity): X_aug = [X_hf | mean_1(X_hf)], k1*k2+k3     -> ARD recipe            1 + 6 runs
  predict:                 X*_aug = [X* | mean_1(X*)]                  -> mean, variance at N* points
following src/MFDataFusion.py:75-100,141-156 and src/abstractMFGP.py:82-106,131-137 of the reference.
Optimiser trajectories are not comparable across back-ends, so every L-BFGS-B run gets the same FIXED
objective+gradient evaluation budget (--evals, default 20; SURVEY.md 8(d) "fixed-budget fit"), enforced EXACTLY
(`eval_cap`): with scipy's `maxfun` alone a run overshoots by up to a line search, and the total then moves by
+-8 % with the last bits of the arithmetic (173 vs 188 evaluations between two versions of one kernel).

  python bench.py [--gpus N --steps K --warmup W]      (N > 1: launched by torch.distributed.run, one rank per GPU)

N > 1 is STRONG scaling of the same job: the randomized restarts and the N* predictive rows are sharded over the
ranks (sharding.py); the LF run is replicated; rank 0 alone runs the sequential first HF run -> restart 0.
Prints ONE JSON line on rank 0.  `value` = milliseconds per fit+predict (lower is better).
The CPU comparator (`cpu_baseline`) is the numpy/LAPACK oracle timed on the host cores of the same box
on a bounded sample (one objective+gradient evaluation per level + the predict products), scaled by the
number of evaluations the GPU path actually issued.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X vendor fp64 peak (matrix = vector); the guide lists no fp64 row
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def hf_4d(x):
    return (np.prod(np.sin(np.pi * x[:, :4]), axis=1) + 5.0)[:, None]


def lf_4d(x):
    return hf_4d(x) - 0.25 * (np.sin(x[:, 0] * np.pi * 0.1) + np.sin(x[:, 1] * np.pi * 0.05)
                              + np.sin(x[:, 2] * 0.15 * np.pi) + np.sin(x[:, 3] * 0.2 * np.pi))[:, None]


def make_data(n_lf, n_hf, n_star, seed=2):
    rng = np.random.default_rng(seed)
    X_lf = rng.uniform(size=(n_lf, 4))
    X_hf = rng.uniform(size=(n_hf, 4))
    X_st = rng.uniform(size=(n_star, 4))
    return X_lf, lf_4d(X_lf), X_hf, X_st


def one_step(args, comm, engines, data):
    """one fit+predict of the 2-fidelity NARGP; returns (mean, var, model)"""
    from multifidelity_datafusion_gps_amd import NARGP
    X_lf, Y_lf, X_hf, X_st = data

    class BudgetNARGP(NARGP):
        lf_max_iters = args.evals
        first_run_max_iters = args.evals
        restart_max_iters = args.evals
        num_restarts = args.restarts
        restart_concurrency = args.concurrency
        eval_cap = args.evals          # exact: scipy's maxfun alone lets a run overshoot by a line search

    model = BudgetNARGP(4, f_exact=hf_4d, f_low=None, lf_X=X_lf, lf_Y=Y_lf, seed=args.seed, comm=comm,
                        engines=engines)
    model.fit(X_hf)
    mean, var = model.predict(X_st)
    return mean, var, model


def cpu_baseline(args, data, n_lf_evals, n_hf_evals):
    """the oracle (numpy + LAPACK dpotrf/dpotri/dpotrs = the routines GPy calls) on the host cores"""
    from oracle import gp_oracle as orc
    X_lf, Y_lf, X_hf, X_st = data
    t = {}
    parts_lf = [(orc.RBF, 0, 4, 0)]
    th_lf, nz_lf = np.array([1.0, 1.0]), 1.0
    t0 = time.perf_counter()
    st_lf = orc.inference(parts_lf, th_lf, nz_lf, X_lf, Y_lf, want_grad=True)
    t["lf_eval_s"] = time.perf_counter() - t0
    # LF posterior mean at X_hf and X* (mean only)
    t0 = time.perf_counter()
    aug_hf = orc.cov(parts_lf, th_lf, X_lf, X_hf).T.dot(st_lf["alpha"])
    aug_st = orc.cov(parts_lf, th_lf, X_lf, X_st).T.dot(st_lf["alpha"])
    t["lf_means_s"] = time.perf_counter() - t0
    Xa = np.hstack([X_hf, aug_hf[:, None]])
    Xsa = np.hstack([X_st, aug_st[:, None]])
    parts_hf = [(orc.RBF, 4, 5, 0), (orc.RBF, 0, 4, 0), (orc.RBF, 0, 4, 1)]
    th_hf = np.ones(6)
    Y_hf = hf_4d(X_hf)
    nz_hf = 0.01 * Y_hf.var()
    t0 = time.perf_counter()
    st_hf = orc.inference(parts_hf, th_hf, nz_hf, Xa, Y_hf, want_grad=True)
    t["hf_eval_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.predict(parts_hf, th_hf, nz_hf, Xa, st_hf, Xsa)
    t["hf_predict_s"] = time.perf_counter() - t0
    total_s = n_lf_evals * t["lf_eval_s"] + n_hf_evals * t["hf_eval_s"] + t["lf_means_s"] + t["hf_predict_s"]
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:  # noqa: BLE001
        cores = os.cpu_count() or 1
    measured = sum(t.values())
    return {"value": round(total_s * 1e3, 1), "unit": "ms", "cores": int(cores), "kind": "port",
            "sample": "oracle (numpy+LAPACK) timed once each at full size: 1 LF objective+gradient eval (%.2fs), "
                      "1 HF eval (%.2fs), LF means (%.2fs), HF predict (%.2fs) = %.1fs measured; scaled to "
                      "%d LF + %d HF evaluations + predicts" % (t["lf_eval_s"], t["hf_eval_s"], t["lf_means_s"],
                                                                 t["hf_predict_s"], measured, n_lf_evals, n_hf_evals)}


def pmc_traffic(kernel, n):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/r01_pmc.json), or None"""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc.json")))
        return int(d[kernel]["traffic_bytes"]) if int(d.get("n", -1)) == int(n) else None
    except Exception:  # noqa: BLE001
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", dest="n", type=int, default=8192, help="N_lf = N_hf = N*")
    ap.add_argument("--evals", type=int, default=20, help="objective evaluations per L-BFGS-B run")
    ap.add_argument("--restarts", type=int, default=6)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--concurrency", type=int, default=2,
                    help="randomized restarts in flight beside the main run (auxiliary engine handles per rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (gloo: CPU rehearsal)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (with --backend gloo) on a one-GPU box")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.single_device:
        local_rank = 0
    import torch  # plumbing only: barrier + device sync + the tiny all-gathers of sharding.TorchComm
    torch.cuda.set_device(local_rank)
    from multifidelity_datafusion_gps_amd import sharding
    from multifidelity_datafusion_gps_amd._lib import Engine
    force_dist = os.environ.get("MFGP_BENCH_FORCE_DIST") == "1"   # rehearsal: a 1-rank process group through the N > 1 code
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            comm = sharding.TorchComm(device="cuda:%d" % local_rank)
        else:
            dist.init_process_group(backend=args.backend)
            comm = sharding.TorchComm(device="cpu")
    else:
        comm = sharding.LocalComm()

    if force_dist:   # the collectives the sharded path uses, through the real backend
        assert comm.allgather_object({"rank": rank}) == [{"rank": r} for r in range(comm.size)]
        got = comm.allgather_rows(np.full((rank + 2, 2), float(rank)))
        assert got.shape[1] == 2 and got[:2].sum() == 0.0

    def barrier():
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    data = make_data(args.n, args.n, args.n)
    engines = {"lf": Engine(local_rank), "hf": Engine(local_rank)}
    for j in range(1, args.concurrency + 1 if args.concurrency > 1 else 1):
        engines["hf#%d" % j] = Engine(local_rank)

    for _ in range(args.warmup):
        one_step(args, comm, engines, data)
    for e in engines.values():
        e.counters(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mean, var, model = one_step(args, comm, engines, data)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1 or force_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=("cuda:%d" % local_rank) if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt * 1e3 / args.steps

    # outside the timed region: the row-block K build + RCCL all-gather layout of north_star,
    # one evaluation each way on the HF level, reported next to the local build it competes with
    rowblock = None
    if world > 1 or force_dist:
        try:
            th, nz = np.ones(6), 0.05
            e = engines["hf"]
            e.eval(th, nz)
            barrier()
            t1 = time.perf_counter()
            f_loc = e.eval(th, nz, want_grad=False)
            barrier()
            t2 = time.perf_counter()
            f_rb = sharding.eval_rowblock_allgather(e, comm, th, nz, want_grad=False)
            barrier()
            t3 = time.perf_counter()
            rowblock = {"local_build_eval_ms": round((t2 - t1) * 1e3, 3), "rowblock_allgather_eval_ms": round((t3 - t2) * 1e3, 3),
                        "nlml_equal": bool(f_loc == f_rb)}
        except Exception as ex:  # noqa: BLE001 - diagnostic only, never fails the bench line
            rowblock = {"error": repr(ex)[:200]}

    if rank == 0:
        clf = engines["lf"].counters()
        chf = engines["hf"].counters()
        for k_, e_ in engines.items():
            if k_.startswith("hf#"):
                for kk, vv in e_.counters().items():
                    chf[kk] += vv
        kinv_ms = clf["kinv_ms"] + chf["kinv_ms"]
        kinv_launches = clf["grad_evals"] + chf["grad_evals"]
        kinv_flops = clf["kinv_flops"] + chf["kinv_flops"]
        kb_ms = clf["kbuild_ms"] + chf["kbuild_ms"]
        kb_bytes = clf["kbuild_bytes"] + chf["kbuild_bytes"]
        evals = clf["evals"] + chf["evals"]
        ach_tf = kinv_flops / (kinv_ms * 1e-3) / 1e12 if kinv_ms > 0 else 0.0
        # the LF level's launches run alone on the GPU (its single L-BFGS-B run is sequential); the HF level's
        # launches share the GPU with the concurrent restarts, which stretches each launch
        ach_tf_alone = clf["kinv_flops"] / (clf["kinv_ms"] * 1e-3) / 1e12 if clf["kinv_ms"] > 0 else 0.0
        ach_gbs = kb_bytes / (kb_ms * 1e-3) / 1e9 if kb_ms > 0 else 0.0
        gpu_eval_ms = (clf["total_ms"] + chf["total_ms"]) / max(evals, 1)
        out = {
            "metric": "gp_fit_predict_wall_ms", "value": round(ms_per_step, 2), "unit": "ms", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2),
            "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "2-fidelity NARGP (data-driven LF GP + composite-kernel HF GP), d=4, "
                                   "N_lf=N_hf=N*=%d, fp64; %d objective+gradient evaluations per L-BFGS-B run, "
                                   "1 LF run + (1+%d) HF runs, then predict" % (args.n, args.evals, args.restarts),
                       "n": args.n, "evals_per_run": args.evals, "restarts": args.restarts,
                       "evals_issued_rank0_per_step": evals / args.steps,
                       "gpu_ms_per_evaluation": round(gpu_eval_ms, 3),
                       "restart_concurrency": args.concurrency,
                       "sharding": "randomized restarts + predictive rows over ranks; LF run replicated; first HF run -> restart 0 on rank 0 only"},
            "roofline": {"kernel": "mfgp_kinv_syrk_f64 (K^-1 = L^-T L^-1, one launch per evaluation)",
                         "bound": "mfma", "achieved": round(ach_tf, 2), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach_tf / FP64_PEAK_TFLOPS, 4),
                         "traffic": pmc_traffic("mfgp_kinv_syrk_f64", args.n),
                         "launches": int(kinv_launches), "avg_launch_ms": round(kinv_ms / max(kinv_launches, 1), 4),
                         "uncontended": {"achieved": round(ach_tf_alone, 2), "frac": round(ach_tf_alone / FP64_PEAK_TFLOPS, 4),
                                         "launches": int(clf["grad_evals"]),
                                         "note": "same kernel, the launches of the LF level only: they run alone on the "
                                                 "GPU, the HF-level launches overlap with %d concurrent restarts"
                                                 % max(args.concurrency, 0)}},
            "roofline_kbuild": {"kernel": "mfgp_kbuild_f64<MODE_TRI> (K(X,X)+noise lower triangle)", "bound": "hbm",
                                "achieved": round(ach_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(ach_gbs / HBM_PEAK_GBS, 4),
                                "traffic": pmc_traffic("mfgp_kbuild_f64<0>", args.n),
                                "launches": int(evals), "avg_launch_ms": round(kb_ms / max(evals, 1), 4)},
            "stage_ms_per_evaluation": {k: round((clf[k] + chf[k]) / max(evals, 1), 4)
                                        for k in ("kbuild_ms", "cholinv_ms", "solve_ms", "kinv_ms", "grad_ms")},
            "result_checksum": {"mean_sum": float(np.sum(mean)), "var_sum": float(np.sum(var))},
        }
        if rowblock is not None:
            out["rowblock_allgather"] = rowblock
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args, data, clf["evals"] / args.steps, chf["evals"] / args.steps)
            out["cpu_baseline"] = cb
            out["config"]["gpu_over_cpu"] = round(cb["value"] / ms_per_step, 2)
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
