"""Find the operand / result layout of v_mfma_f64_4x4x4_4b_f64 on the GPU, and what CBSZ / ABID do to it (GPU box).
Random integer-valued operands (exact arithmetic), every candidate lane map tried; prints the maps that reproduce the
device's results for all seven (CBSZ, ABID) settings."""
import ctypes, itertools, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libmfgp_probes.so"))
dp = ctypes.POINTER(ctypes.c_double)
lib.mfgp_probe_mfma444_layout.restype = ctypes.c_int32
lib.mfgp_probe_mfma444_layout.argtypes = [ctypes.c_int32, dp, dp, dp, dp]
rng = np.random.default_rng(0)
a = rng.integers(-8, 9, 64).astype(float); b = rng.integers(-8, 9, 64).astype(float); c = rng.integers(-8, 9, 64).astype(float)
out = np.zeros((7, 64))
rc = lib.mfgp_probe_mfma444_layout(0, a.ctypes.data_as(dp), b.ctypes.data_as(dp), c.ctypes.data_as(dp), out.ctypes.data_as(dp))
assert rc == 0, rc
settings = [(0, 0), (1, 0), (1, 1), (2, 0), (2, 1), (2, 2), (2, 3)]
# lane l = 16*blk + 4*hi + lo.  Candidate maps: A element (i, k) of block blk sits in lane 16*blk + 4*k + i ("ki") or 4*i + k ("ik");
# B element (k, j): 4*k + j ("kj") or 4*j + k ("jk"); D element (i, j): 4*i + j ("ij") or 4*j + i ("ji").
def lane(blk, hi, lo): return 16 * blk + 4 * hi + lo
found = []
for am, bm, dm in itertools.product(("ki", "ik"), ("kj", "jk"), ("ij", "ji")):
    for bcast in ("group", "none"):
        ok = True
        for s_idx, (cbsz, abid) in enumerate(settings):
            exp = np.zeros(64)
            for blk in range(4):
                ablk = blk
                if bcast == "group" and cbsz:
                    g = 1 << cbsz
                    ablk = (blk // g) * g + abid
                for i in range(4):
                    for j in range(4):
                        ld = lane(blk, i, j) if dm == "ij" else lane(blk, j, i)
                        acc = c[ld]
                        for k in range(4):
                            la = lane(ablk, k, i) if am == "ki" else lane(ablk, i, k)
                            lb = lane(blk, k, j) if bm == "kj" else lane(blk, j, k)
                            acc += a[la] * b[lb]
                        exp[ld] = acc
            if not np.array_equal(exp, out[s_idx]):
                ok = False
                if s_idx == 0:
                    break
        if ok:
            found.append((am, bm, dm, bcast))
        elif bcast == "none":
            pass
print("maps reproducing ALL settings:", found)
for am, bm, dm in itertools.product(("ki", "ik"), ("kj", "jk"), ("ij", "ji")):   # which maps fit the plain (0,0) setting
    exp = np.zeros(64)
    for blk in range(4):
        for i in range(4):
            for j in range(4):
                ld = lane(blk, i, j) if dm == "ij" else lane(blk, j, i)
                exp[ld] = c[ld] + sum(a[lane(blk, k, i) if am == "ki" else lane(blk, i, k)] * b[lane(blk, k, j) if bm == "kj" else lane(blk, j, k)] for k in range(4))
    if np.array_equal(exp, out[0]):
        print("plain setting fits:", am, bm, dm)
print("identical to the plain result:", [settings[i] for i in range(7) if np.array_equal(out[i], out[0])])
np.save(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "mfma444_layout.npy"), np.vstack([a, b, c, out]))
