"""Which CUs does a hipExtStreamCreateWithCUMask mask leave to a queue on this part (8 XCCs x 4 SEs x 8-9 CUs)?  A grid of 512
64-KB workgroups (two per CU when all 256 CUs are available) that stay resident ~300 us records (XCC, SE, CU) per workgroup.
usage: cumask.py"""
import collections
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libmfgp_probes.so"))
u32p = ctypes.POINTER(ctypes.c_uint32)
lib.mfgp_probe_placement_masked.restype = ctypes.c_int32
lib.mfgp_probe_placement_masked.argtypes = [ctypes.c_int32] * 5 + [u32p, u32p, ctypes.POINTER(ctypes.c_double)]


def run(name, bits, G=512, lds=65536):
    nwords = (len(bits) + 31) // 32
    mask = np.zeros(nwords, dtype=np.uint32)
    for i, b in enumerate(bits):
        if b:
            mask[i // 32] |= np.uint32(1 << (i % 32))
    out = np.zeros(2 * G, dtype=np.uint32)
    ms = ctypes.c_double()
    rc = lib.mfgp_probe_placement_masked(0, G, lds, 300, nwords, mask.ctypes.data_as(u32p), out.ctypes.data_as(u32p), ctypes.byref(ms))
    if rc:
        print("%-34s rc %d" % (name, rc))
        return
    hw, xcc = out[0::2], out[1::2] & 0xF
    cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7
    per_cu = collections.Counter(zip(xcc.tolist(), se.tolist(), cu.tolist()))
    per_xcc = collections.Counter(k[0] for k in per_cu)
    per_se = collections.Counter((k[0], k[1]) for k in per_cu)
    print("%-34s bits set %3d/%3d: %3d distinct CUs used; CUs per XCC %s; launch %.2f ms; CUs per (xcc0, se) %s" % (
        name, int(sum(bits)), len(bits), len(per_cu), [per_xcc.get(x, 0) for x in range(8)], ms.value,
        [per_se.get((0, s), 0) for s in range(4)]), flush=True)
    return per_cu


full = run("all 256 bits", [1] * 256)
allc = set(full) if full else set()
for name, bits in (
        ("bit 0 cleared", [0] + [1] * 255),
        ("bits 0-7 cleared", [0] * 8 + [1] * 248),
        ("bits 0-31 cleared", [0] * 32 + [1] * 224),
        ("every 32nd bit cleared", [0 if i % 32 == 0 else 1 for i in range(256)]),
        ("every 8th bit cleared", [0 if i % 8 == 0 else 1 for i in range(256)]),
        ("bits 248-255 cleared", [1] * 248 + [0] * 8),
        ("only bits 0-31 set", [1] * 32 + [0] * 224),
        ("only bits 0-7 set", [1] * 8 + [0] * 248),
        ("only every 8th bit set", [1 if i % 8 == 0 else 0 for i in range(256)]),
        ("288 bits all set", [1] * 288),
        ("288 bits, 0-7 cleared", [0] * 8 + [1] * 280),
):
    got = run(name, bits)
    if got is not None and allc:
        missing = sorted(allc - set(got))
        print("      CUs not used (xcc, se, cu):", missing[:40], "..." if len(missing) > 40 else "")
