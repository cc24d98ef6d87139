"""Child program of tests/test_bench_launcher.py: what one rank of a `bench.py --gpus N` job sees.  Started by
bench.launch_ranks; joins the rendezvous from the environment the launcher exported, runs the sharded fit + predict with
the tests-only oracle double as the engine, and rank 0 prints one JSON line.  argv: [fail_rank] -- that rank exits 7
after the rendezvous (the launcher must stop the others and report a non-zero code)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from multifidelity_datafusion_gps_amd import sharding  # noqa: E402
from tests.test_sharding_gloo import _run_model  # noqa: E402

fail_rank = int(sys.argv[1]) if len(sys.argv) > 1 else -1
comm = sharding.comm_from_env(timeout=60)
rank = int(os.environ["RANK"])
seen = comm.allgather_object({"rank": rank, "local_rank": int(os.environ["LOCAL_RANK"]), "world": int(os.environ["WORLD_SIZE"]),
                              "token": bool(os.environ.get("MFGP_COMM_TOKEN")), "addr": os.environ["MASTER_ADDR"]})
if rank == fail_rank:
    sys.exit(7)
try:
    res = _run_model(comm, 2)
    comm.barrier()      # with a failed rank the others wait here until the launcher stops them, or lose their peer
except ConnectionError:
    sys.exit(4)         # bench.EXIT_PEER_LOST, as bench.py's own ranks do
evals = comm.allgather_object(int(res["evals"]))     # high-fidelity evaluations each rank issued (its restarts, the chain on rank 0)
if rank == 0:
    print(json.dumps({"seen": seen, "mean_sum": float(res["mean"].sum()), "var_sum": float(res["var"].sum()),
                      "theta": res["theta"].tolist(), "hf_evals_per_rank": evals}), flush=True)
comm.close()
