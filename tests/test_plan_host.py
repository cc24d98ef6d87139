"""CPU checks of the factorisation planner (multifidelity_datafusion_gps_amd/csrc/plan.cpp) -- no GPU needed.

tests/host_plan/plan_sim.cpp compiles the planner itself with g++ and, for the plans it emits,
  * executes every step in enqueue order with a plain-C restatement of the tile-GEMM task and of the leaf, on matrices
    pre-filled with NaN wherever nothing has been written, and checks L L^T = A, X L = I, the mirrored storage of X and
    K^-1 = X^T X;
  * checks the two-stream schedule for data races (vector clocks over the streams, 64x64 cells, tasks of one launch
    concurrent), that every event is recorded before it is waited for, and that the bulk stream is joined at the end.
The checker checks itself on three mutated plans (a dropped wait must be reported)."""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "tests", "host_plan", "plan_sim.cpp"),
       os.path.join(ROOT, "multifidelity_datafusion_gps_amd", "csrc", "plan.cpp")]
LIB = os.path.join(ROOT, "tests", "host_plan", "libplan_sim.so")

_DRIVER = r"""
import ctypes, json, sys
lib = ctypes.CDLL(sys.argv[1])
lib.plan_sim.restype = ctypes.c_int
lib.plan_sim.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_double), ctypes.c_char_p, ctypes.c_int]
out = []
for spec in json.loads(sys.argv[2]):
    rep = (ctypes.c_double * 8)()
    msg = ctypes.create_string_buffer(512)
    rc = lib.plan_sim(*spec, rep, msg, 512)
    out.append([rc, list(rep), msg.value.decode()])
print(json.dumps(out))
"""


@pytest.fixture(scope="module")
def simlib():
    deps = SRC + [os.path.join(ROOT, "multifidelity_datafusion_gps_amd", "csrc", "plan.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", LIB] + SRC, check=True)
    return LIB


_WORKERS = max(1, min(4, (os.cpu_count() or 2) // 2))      # checker processes side by side (each is single-threaded)


def _spread(specs, call):
    """`call(list of specs) -> list of results`, the specs dealt round-robin over _WORKERS concurrent calls; results in spec order"""
    from concurrent.futures import ThreadPoolExecutor
    specs = list(specs)
    parts = [specs[k::_WORKERS] for k in range(_WORKERS) if specs[k::_WORKERS]]
    with ThreadPoolExecutor(len(parts) or 1) as pool:
        outs = list(pool.map(call, parts))
    res = [None] * len(specs)
    for k, out in enumerate(outs):
        res[k::_WORKERS] = out
    return res


def _run(lib, specs, env=None):
    """the planner reads its switches from the environment when it plans: subprocesses per environment"""
    import json
    e = dict(os.environ)
    e.update(env or {})

    def call(part):
        r = subprocess.run([sys.executable, "-c", _DRIVER, lib, json.dumps(part)], env=e, capture_output=True, text=True,
                           check=True)
        return json.loads(r.stdout)
    return _spread(specs, call)


ENVS = [{}, {"MFGP_MACRO": "2"}, {"MFGP_MACRO": "3", "MFGP_SHIFT": "0"}, {"MFGP_SHIFT": "0"}, {"MFGP_KINV_STREAM": "0"},
        {"MFGP_T128_MIN": "10"}, {"MFGP_MACRO": "1"}, {"MFGP_MACRO": "4", "MFGP_SHIFT": "1", "MFGP_CHAIN_SLIM": "1"},
        {"MFGP_MACRO": "2", "MFGP_KINV_STREAM": "0", "MFGP_SHIFT": "0"},
        {"MFGP_PLAN": "levels"}, {"MFGP_PLAN": "levels", "MFGP_MACRO": "2", "MFGP_SHIFT": "0"}, {"MFGP_PLAN": "recursive"}]


@pytest.mark.parametrize("env", ENVS, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()) or "default")
def test_plans_factorise_invert_and_accumulate_kinv_without_races(simlib, env):
    # (nblk, numeric, want_grad, slack rows of capacity, mutate)
    specs = [(nb, 1, g, 128 if nb % 2 else 0, 0) for nb in (1, 2, 3, 5, 8, 9) for g in (0, 1)]
    if not env:   # the largest single-macro plan of the defaults (13 block columns, K^-1 on the chain)
        specs += [(13, 1, g, 0, 0) for g in (0, 1)]
    for (nb, _, g, _, _), (rc, rep, msg) in zip(specs, _run(simlib, specs, env)):
        assert rc == 0, (nb, g, msg)
        assert rep[0] < 1e-14 and rep[1] < 1e-12 and rep[2] == 0.0, (nb, g, rep)     # L L^T = A, X L = I, S mirrored
        if g:
            assert rep[3] < 1e-12, (nb, rep)                                            # K^-1 = X^T X
        assert rep[4] == 0


@pytest.mark.parametrize("env", [{}, {"MFGP_SHIFT": "1"}, {"MFGP_SHIFT": "0"}, {"MFGP_MACRO": "6"}, {"MFGP_MACRO": "8", "MFGP_SHIFT": "0"},
                                 {"MFGP_PLAN": "levels"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()) or "default")
def test_schedules_of_the_bench_sizes_are_race_free(simlib, env):
    """race check only (no arithmetic) at the block counts of the BASELINE configurations: 32 (N = 4096), 47 / 48 / 49
    (the slim-chain switch), 64 (N = 8192), 128 (N = 16384), and either side of every size at which the planner's defaults
    change (macro panel length, merged column launch, tile sizes)"""
    specs = [(nb, 0, 1, 0, 0) for nb in (8, 9, 13, 14, 16, 24, 25, 32, 33, 40, 41, 47, 48, 49, 51, 52, 55, 56, 63, 64, 128)]
    for spec, (rc, rep, msg) in zip(specs, _run(simlib, specs, env)):
        assert rc == 0 and rep[4] == 0, (spec, msg)


@pytest.mark.parametrize("div", [3, 5, 9])
def test_batch_plans_are_race_free_and_factorise(simlib, div):
    """the plan of a BATCHED pass (mfgp_eval_batch: tile sizes chosen for `div` times the workgroups per launch -- 128-tiles sooner,
    64-tile chain steps from 32 block columns) through the same checker: arithmetic at small block counts, races at the block counts
    where a batch plan differs from the single one"""
    env = {"PLAN_SIM_BATCH_DIV": str(div)}
    specs = [(nb, 1, g, 0, 0) for nb in (3, 8, 13) for g in (0, 1)]
    for (nb, _, g, _, _), (rc, rep, msg) in zip(specs, _run(simlib, specs, env)):
        assert rc == 0, (nb, g, msg)
        assert rep[0] < 1e-14 and rep[1] < 1e-12 and rep[2] == 0.0, (nb, g, rep)
    specs = [(nb, 0, 1, 0, 0) for nb in (14, 24, 31, 32, 33, 40, 47, 48, 56, 64)]
    for spec, (rc, rep, msg) in zip(specs, _run(simlib, specs, env)):
        assert rc == 0 and rep[4] == 0, (spec, msg)


def test_the_checker_catches_a_dropped_dependency(simlib):
    ok, no_chain_wait, no_join, no_leaf_wait = _run(simlib, [(16, 0, 1, 0, m) for m in (0, 1, 2, 3)])
    assert ok[0] == 0
    assert no_chain_wait[0] == 1 and no_chain_wait[1][4] > 0 and "race" in no_chain_wait[2]
    assert no_join[0] == -3 and "not joined" in no_join[2]
    assert no_leaf_wait[0] == 1 and no_leaf_wait[1][4] > 0


_SHARD_DRIVER = r"""
import ctypes, json, sys
lib = ctypes.CDLL(sys.argv[1])
fn = getattr(lib, sys.argv[3] if len(sys.argv) > 3 else "plan_sim_sharded")
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_char_p, ctypes.c_int]
out = []
for nb, size in json.loads(sys.argv[2]):
    rep = (ctypes.c_double * 8)()
    msg = ctypes.create_string_buffer(512)
    rc = fn(nb, size, rep, msg, 512)
    out.append([rc, list(rep), msg.value.decode()])
print(json.dumps(out))
"""


def _run_sharded(lib, specs, env, entry="plan_sim_sharded"):
    import json
    e = dict(os.environ)
    e.update(env)

    def call(part):
        r = subprocess.run([sys.executable, "-c", _SHARD_DRIVER, lib, json.dumps(part), entry], env=e, capture_output=True, text=True,
                           check=True)
        return json.loads(r.stdout)
    return _spread(specs, call)


@pytest.mark.parametrize("env", [{}, {"MFGP_MACRO": "2"}, {"MFGP_MACRO": "3", "MFGP_SHIFT": "0"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()) or "default")
def test_sharded_plans_reproduce_the_single_plan_bit_for_bit(simlib, env):
    """mfgp_eval_sharded on the CPU: every rank's plan (the Cholesky in full, the work on the rows of L^-T / K^-1 for its own
    128-row blocks) executed on its own copy of the matrices, the rows of X^T exchanged from their owners, the lower part
    rebuilt from the upper one, every rank's K^-1 rows accumulated -- each rank's plan race-free, and L, S and the owned rows of
    K^-1 BITWISE what one rank computes alone (same tasks, same arithmetic); the largest rank plan shrinks towards
    1/3 + 2/(3 G) of the single plan's tasks as the matrix grows."""
    # (nblk, ranks); 14 blocks = N 1792: the first size with more than one macro panel.  The full list under the default switches,
    # three of them under the others (the checker executes every rank's plan on the CPU: seconds per rank at 14 blocks)
    # (16 blocks on 4 ranks and 9 on 4 run in the distributed-Cholesky test below, which shares every B / X^T / K^-1 task with this plan)
    specs = [(3, 2), (5, 2), (8, 3), (14, 3)] if not env else [(5, 2), (9, 4), (14, 2)]
    results = _run_sharded(simlib, specs, env)
    for (nb, size), (rc, rep, msg) in zip(specs, results):
        assert rc == 0 and rep[0] == 0, (nb, size, msg)
        assert rep[1] < 1e-12, (nb, size, rep)               # X L = I after the exchange
        assert rep[2] == 0 and rep[3] == 0, (nb, size, rep)    # no word of L / S / own K^-1 rows differs from the single run
        assert rep[4] <= 1.0


def _macro_columns(nb, env=None):
    """plan.cpp sweep_macro_columns: block columns per macro panel of the sweep"""
    m = int((env or {}).get("MFGP_MACRO", 0))
    return m if m > 0 else (5 if nb >= 80 else (4 if nb >= 56 else (2 if nb >= 25 else (3 if nb > 13 else nb))))


def _dist_collectives(nb, env=None):
    """plan.cpp dist_collectives: one all-gather per block column but the last + one broadcast per macro panel (the diagonal block of a
    macro panel's first column), or round 5's 2 nb - 1 under MFGP_DIST_FUSE=0"""
    if (env or {}).get("MFGP_DIST_FUSE") == "0":
        return 2 * nb - 1
    mb = _macro_columns(nb, env)
    return nb - 1 + (nb + mb - 1) // mb


@pytest.mark.parametrize("env", [{}, {"MFGP_MACRO": "2"}, {"MFGP_MACRO": "3", "MFGP_SHIFT": "0"}, {"MFGP_DIST_FUSE": "0"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()) or "default")
def test_distributed_cholesky_plans_reproduce_the_single_plan_bit_for_bit(simlib, env):
    """SURVEY 8(e) "Cholesky" (round 5): the 1-D block-cyclic factorisation over a rank group (plan.h Shard::dist).  Every rank's plan
    holds only ITS rows of the panels and of the trailing updates of A (the leaf of the diagonal blocks it owns), plus two exchange
    steps per block column -- the diagonal blocks from their owner, the panel column from the owners of its rows -- executed here
    in lock step over the ranks' own copies of the matrices (pre-filled with NaN: a rank that read a row it never received would
    poison its result).  Each rank's schedule is race-free, the exchange steps come in the same order on every rank, L, S and
    the owned rows of K^-1 are BITWISE the single plan's, and the largest rank's task count falls towards 1/G of it.
    Round 6: ONE exchange per column wherever the next column belongs to the same macro panel -- its owner brings the diagonal
    block up to date from its own row of the panel, factorises it before the all-gather, and the all-gather carries it along
    (Step::carry): nb - 1 + ceil(nb / MB) collectives instead of 2 nb - 1 (MFGP_DIST_FUSE=0: round 5's form), the same bytes."""
    specs = [(3, 2), (8, 3), (14, 3), (16, 4)] if not env else [(5, 2), (9, 4), (14, 2)]
    results = _run_sharded(simlib, specs, env, "plan_sim_dist")
    for (nb, size), (rc, rep, msg) in zip(specs, results):
        assert rc == 0 and rep[0] == 0, (nb, size, rc, msg)
        assert rep[1] < 1e-12, (nb, size, rep)               # X L = I on rank 0 (its L is part computed, part received)
        assert rep[2] == 0 and rep[3] == 0, (nb, size, rep)    # no word of L / S / own K^-1 rows differs from the single run
        assert rep[4] <= 1.0
        # bytes through the Cholesky's exchange steps: 2 diagonal blocks + the blocks below, per block column
        assert rep[5] == 8 * 128 * 128 * sum(2 + (nb - 1 - c) for c in range(nb)), (nb, size, rep[5])
        assert rep[6] == _dist_collectives(nb, env), (nb, size, rep[6], _dist_collectives(nb, env))
        carried = 0 if env.get("MFGP_DIST_FUSE") == "0" else nb - (nb + _macro_columns(nb, env) - 1) // _macro_columns(nb, env)
        assert rep[7] == carried, (nb, size, rep[7], carried)
    if not env:
        share = {(nb, size): rep[4] for (nb, size), (rc, rep, msg) in zip(specs, results)}
        assert share[(16, 4)] < 0.45 and share[(14, 3)] < 0.55, share   # (sharded without dist: 1/3 + 2/(3 G) = 0.5 / 0.56 at best)


_SCHED_DRIVER = r"""
import ctypes, json, sys
lib = ctypes.CDLL(sys.argv[1])
fn = lib.plan_sim_group_schedule
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_double), ctypes.c_char_p, ctypes.c_int]
out = []
for nb, size, dist in json.loads(sys.argv[2]):
    rep = (ctypes.c_double * 8)()
    msg = ctypes.create_string_buffer(512)
    rc = fn(nb, size, dist, rep, msg, 512)
    out.append([rc, list(rep), msg.value.decode()])
print(json.dumps(out))
"""


def test_group_schedules_at_the_sizes_of_the_eight_rank_job(simlib):
    """What no one-GPU rig can run: the plans of an 8-rank group at the sizes they are made for -- the bench's LF group (64 block
    columns on 8 ranks) and chain group (on 3), and the distributed Cholesky at N = 8192 / 16384 (its default threshold) / 32768 on 8
    ranks.  Schedule only (no arithmetic): every rank's plan free of races with every wait behind its record, the exchange steps met
    in the same order by all ranks (nb - 1 + ceil(nb / MB) of them in the one-exchange-per-column form, 8 * 128 * 128 * (2 + blocks
    below) bytes per column either way), and the largest rank's task list near 1/8 of the single plan's under the distributed Cholesky."""
    import json
    specs = [(64, 8, 0), (64, 3, 0), (64, 8, 1), (128, 8, 1), (128, 8, 0), (256, 8, 1), (128, 5, 1)]

    def call(part):
        r = subprocess.run([sys.executable, "-c", _SCHED_DRIVER, simlib, json.dumps(part)], capture_output=True, text=True, check=True)
        return json.loads(r.stdout)
    for (nb, size, dist), (rc, rep, msg) in zip(specs, _spread(specs, call)):
        assert rc == 0 and rep[0] == 0, (nb, size, dist, rc, msg)
        if dist:
            assert rep[6] == _dist_collectives(nb), (nb, size, rep)
            assert rep[5] == 8 * 128 * 128 * sum(2 + (nb - 1 - c) for c in range(nb)), (nb, size, rep)
            assert rep[4] < 1.0 / size + 0.1, (nb, size, rep)
        else:
            assert rep[6] == 0 and rep[4] < 1.0, (nb, size, rep)


# ---- the same planner + checker under AddressSanitizer / UndefinedBehaviorSanitizer (CPU build) -------------
ASAN_BIN = os.path.join(ROOT, "tests", "host_plan", "plan_sim_asan")
ASAN_SRC = SRC + [os.path.join(ROOT, "tests", "host_plan", "plan_sim_main.cpp")]


@pytest.fixture(scope="module")
def asan_bin():
    deps = ASAN_SRC + [os.path.join(ROOT, "multifidelity_datafusion_gps_amd", "csrc", "plan.h")]
    if not os.path.exists(ASAN_BIN) or any(os.path.getmtime(d) > os.path.getmtime(ASAN_BIN) for d in deps):
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-fno-omit-frame-pointer", "-o", ASAN_BIN] + ASAN_SRC, check=True)
    return ASAN_BIN


@pytest.mark.parametrize("env", [{}, {"MFGP_MACRO": "2"}, {"MFGP_PLAN": "levels"},
                                 {"MFGP_PLAN": "recursive"}, {"MFGP_MACRO": "3", "MFGP_SHIFT": "0", "MFGP_KINV_STREAM": "0"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()) or "default")
def test_planner_and_checker_are_clean_under_asan_and_ubsan(asan_bin, env):
    """heap / stack overflows, use after free, signed overflow, misaligned or out-of-range accesses in plan.cpp while it
    plans and while its plans are executed: numeric plans up to 9 block columns, schedules up to 128 (N = 16384)"""
    specs = [(nb, 1, g, 128 if nb % 2 else 0, 0) for nb in (1, 2, 3, 5, 9) for g in (0, 1)]
    specs += [(nb, 0, 1, 0, 0) for nb in (13, 14, 32, 48, 49, 64, 128)]
    e = dict(os.environ)
    e.update(env)
    e["ASAN_OPTIONS"] = "detect_leaks=1:abort_on_error=0:exitcode=97"
    e["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    def call(part):
        r = subprocess.run([asan_bin] + [str(v) for spec in part for v in spec], env=e, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
        out = r.stdout.strip().splitlines()
        assert len(out) == len(part)
        return out
    lines = _spread(specs, call)
    assert len(lines) == len(specs)
    for spec, line in zip(specs, lines):
        assert line.split()[0] == "0", (spec, line)
