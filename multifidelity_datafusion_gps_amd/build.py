"""Build libmfgp_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

The shared library is the product: there is no CPU fallback.  `python -m
multifidelity_datafusion_gps_amd.build` (or __graft_entry__.build()) cross-compiles it without a GPU.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmfgp_hip.so")
SOURCES = ["gemm_f64.hip", "leaf_f64.hip", "covariance.hip", "vecops.hip", "trimv_f64.hip", "comm_rccl.hip", "plan.cpp", "mfgp_api.hip", "api_batch.hip",
           "api_sharded.hip", "api_predict.hip", "api_debug.hip"]
HEADERS = ["mfgp_internal.h", "api_shared.h", "plan.h", os.path.join("..", "..", "include", "mfgp.h")]


def source_hash():
    """sha256 (first 16 hex digits) over the library's sources -- csrc/*.hip, *.cpp, *.h and include/mfgp.h, in a fixed order:
    embedded into libmfgp_hip.so when it is built (mfgp_build_id) and written into every counter summary under profiles/ when
    it is produced, so that bench.py can tell whether committed counters still describe the library it loaded"""
    import hashlib
    h = hashlib.sha256()
    for rel in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libmfgp_hip.so)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


PROBE_SRC = os.path.join(os.path.dirname(HERE), "tools", "probes", "probes.hip")
PROBE_LIB = os.path.join(os.path.dirname(HERE), "tools", "probes", "libmfgp_probes.so")


def build_probes(force=False, verbose=False):
    """tools/probes/libmfgp_probes.so: hardware probes for tests / tools (NOT part of the product library)"""
    if not os.path.exists(PROBE_SRC):
        return None
    if not force and os.path.exists(PROBE_LIB) and os.path.getmtime(PROBE_LIB) >= os.path.getmtime(PROBE_SRC):
        return PROBE_LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", PROBE_LIB, PROBE_SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libmfgp_probes.so")
    return PROBE_LIB


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    # MFGP_BUILD_DEFINES: extra -D switches of lab builds on the GPU box; the library the repository ships is built without any
    extra = os.environ.get("MFGP_BUILD_DEFINES", "").split()
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-result", "-Wno-unused-value", '-DMFGP_SRC_HASH="%s"' % source_hash()] + extra + ["-o", LIB] + \
          [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libmfgp_hip.so")
    if verbose and (r.stdout or r.stderr):
        print(r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
