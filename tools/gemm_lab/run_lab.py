"""Run the gemm_lab variants on the GPU box: correctness of every real variant against numpy at n = 1024 (all flag
combinations the planner emits), then launch times at n = 8192 for K = 512 / 2048.  usage: run_lab.py [variant substring]"""
import ctypes, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libgemm_lab.so"))
dp = ctypes.POINTER(ctypes.c_double)
lib.lab_variant_name.restype = ctypes.c_char_p
lib.lab_run.restype = ctypes.c_int
lib.lab_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, dp, dp, dp, dp]
names = [lib.lab_variant_name(i).decode() for i in range(lib.lab_num_variants())]
wants = sys.argv[1:] or [""]


def fill(n, seed):
    i = np.arange(1, n * n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = i * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)
        h ^= h >> np.uint64(29); h *= np.uint64(0xBF58476D1CE4E5B9); h ^= h >> np.uint64(32)
    return (((h >> np.uint64(11)).astype(np.float64) / 9007199254740992.0 - 0.5) * 2.0).reshape(n, n)


def reference(n, K, tile, flags, beta):
    S, C0 = fill(n, 1), fill(n, 2)
    out = np.empty((n, n))
    r = np.arange(tile)[:, None]; k = np.arange(K)[None, :]
    ma = np.ones((tile, K)); mb = np.ones((tile, K))
    if flags & 1: ma = ma * (k <= r + K - tile)
    if flags & 2: ma = ma * (k >= r)
    if flags & 4: mb = mb * (k <= r + K - tile)
    if flags & 8: mb = mb * (k >= r)
    for i in range(n // tile):
        A = S[i * tile:(i + 1) * tile, :K] * ma
        for j in range(n // tile):
            B = S[j * tile:(j + 1) * tile, :K] * mb
            out[i * tile:(i + 1) * tile, j * tile:(j + 1) * tile] = beta * C0[i * tile:(i + 1) * tile, j * tile:(j + 1) * tile] + A @ B.T
    return out


ms, md = ctypes.c_double(), ctypes.c_double()
print("== correctness, n = 1024, K = 256 (flags x beta)")
for v, name in enumerate(names):
    if not any(w in name for w in wants) or not name.endswith("m0"):
        continue
    tile = 128 if "128" in name else 64
    worst = 0.0
    for flags in (0, 1, 2, 4, 8, 5, 10, 9, 6):
        for beta in (0.0, 1.0):
            ref = np.ascontiguousarray(reference(1024, 256, tile, flags, beta))
            rc = lib.lab_run(v, 1024, 256, flags, beta, 1, ctypes.byref(ms), ref.ctypes.data_as(dp), None, ctypes.byref(md))
            assert rc == 0, (name, rc)
            worst = max(worst, md.value)
    print("%-14s max |C - numpy| = %.3e %s" % (name, worst, "OK" if worst < 1e-10 else "WRONG"), flush=True)
if hasattr(lib, "lab_set_persist_grid"):
    lib.lab_set_persist_grid(int(os.environ.get("LAB_PERSIST_GRID", "250")))
print("== launch time, n = 8192")
for K in (512, 2048):
    for v, name in enumerate(names):
        if not any(w in name for w in wants):
            continue
        rc = lib.lab_run(v, 8192, K, 0, 1.0, 5, ctypes.byref(ms), None, None, None)
        assert rc == 0, (name, rc)
        fl = 2.0 * 8192 * 8192 * K
        print("K=%-5d %-14s %8.3f ms  %6.1f TFLOP/s" % (K, name, ms.value, fl / ms.value / 1e9), flush=True)
