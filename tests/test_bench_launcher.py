"""`bench.py --gpus N` as the driver may start it -- a plain process -- must become N ranks, and must not succeed
quietly when it cannot.  CPU tests of the launcher: the environment every rank receives, the
result of the sharded job it starts (oracle double as the engine), and the exit code when a rank fails or when the
engine / the GPU is missing."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tests", "launcher", "rank_probe.py")


def _launch(n, argv):
    code = ("import sys; sys.path.insert(0, %r); import bench; sys.exit(bench.launch_ranks(%d, %r, script=%r))"
            % (ROOT, n, list(argv), PROBE))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MFGP_COMM_TOKEN")}
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)


def test_launcher_starts_one_rank_per_gpu_with_the_rendezvous_environment():
    from multifidelity_datafusion_gps_amd.sharding import LocalComm
    from tests.test_sharding_gloo import _run_model
    r = _launch(3, [])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["seen"] == [{"rank": k, "local_rank": k, "world": 3, "token": True, "addr": "127.0.0.1"} for k in range(3)]
    ref = _run_model(LocalComm(), 2)      # the 3-rank job is the single-process job, sharded
    np.testing.assert_allclose(line["theta"], ref["theta"], rtol=1e-12)
    assert abs(line["mean_sum"] - float(ref["mean"].sum())) < 1e-5 and abs(line["var_sum"] - float(ref["var"].sum())) < 1e-5


def test_eight_rank_job_is_the_single_process_job():
    """the size BASELINE.json's metric names (1/2/4/8 GPU), rehearsed on the CPU: 8 ranks over the launcher and the TCP rendezvous
    (a one-GPU box admits at most 6 processes on its card, so the GPU rehearsal stops at 6 ranks: profiles/r05_bench_n6_*).  Restart
    layout of AbstractMFGP.assign_restarts at 8 ranks: the chain (first run -> restart 0) on rank 0, no optimiser run on ranks 1-2,
    one randomized restart each on ranks 3-7; every rank ends with the single process's winner and its share of the predictive rows."""
    from multifidelity_datafusion_gps_amd.sharding import LocalComm
    from tests.test_sharding_gloo import _run_model
    r = _launch(8, [])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert [s["rank"] for s in line["seen"]] == list(range(8)) and all(s["world"] == 8 for s in line["seen"])
    ref = _run_model(LocalComm(), 2)
    np.testing.assert_allclose(line["theta"], ref["theta"], rtol=1e-12)
    assert abs(line["mean_sum"] - float(ref["mean"].sum())) < 1e-5 and abs(line["var_sum"] - float(ref["var"].sum())) < 1e-5
    ev = line["hf_evals_per_rank"]
    # ranks 1-2 adopt the winner (one factorisation at it, perhaps a lazy re-evaluation); ranks 3-7 ran one restart each; rank 0 the chain
    assert max(ev[1:3]) <= 3 and min(ev[3:]) > 3 and ev[0] > max(ev[3:]) and sum(ev) <= ref["evals"] + 3 * 8, ev


def test_launcher_reports_a_failed_rank_and_stops_the_others():
    r = _launch(2, ["1"])
    assert r.returncode == 7
    assert "rank 1 exited with code 7" in r.stderr
    assert r.stdout.strip() == ""         # no result line from a job that lost a rank


def test_bench_gpus_n_is_loud_without_an_engine():
    """no GPU here: every rank's Engine() raises EngineUnavailable -> the N-rank job exits non-zero, prints no line"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--points", "256", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "exited with code" in r.stderr and '"metric"' not in r.stdout


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr
