"""Per-launch work of the factorisation plan (from the planner itself, compiled for the host: tests/host_plan) next to a
rocprofv3 kernel-trace timeline of one evaluation: which launches run the matrix pipe well and which do not.
usage: plan_flops.py <nblk> [timeline.txt from tools/trace_timeline.py]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "host_plan", "libplan_sim.so")
SRC = [os.path.join(ROOT, "tests", "host_plan", "plan_sim.cpp"), os.path.join(ROOT, "multifidelity_datafusion_gps_amd", "csrc", "plan.cpp")]
if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in SRC):
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", LIB] + SRC, check=True)
lib = ctypes.CDLL(LIB)
nb = int(sys.argv[1])
buf = (ctypes.c_double * (7 * 4096))()
n = lib.plan_steps(nb, 1, buf, 4096)
steps = [tuple(buf[7 * i + k] for k in range(7)) for i in range(n)]
durs = {0: [], 1: []}
if len(sys.argv) > 2:
    for line in open(sys.argv[2]):
        f = line.split()
        if not f or "kbuild" in f[0] or "rowdot" in f[0] or "trimv" in f[0] or "grad" in f[0] or "finish" in f[0]:
            continue
        q = 0 if f[1] == "q1" else 1
        durs[q].append((f[0], int(f[4]), float(f[-2])))
idx = {0: 0, 1: 0}
tot = {0: [0.0, 0.0], 1: [0.0, 0.0]}
for strm, kind, tile, cnt, gf, kmax, kmin in steps:
    s = int(strm)
    if kind == 2 or (kind == 1 and cnt == 0):
        continue
    d = None
    if durs[s] and idx[s] < len(durs[s]):
        d = durs[s][idx[s]]; idx[s] += 1
    name = "leaf" if kind == 0 else "gemm%d" % tile
    line = "%s %-8s tasks %5d  %8.2f Gflop  K %5d..%-5d" % ("main" if s == 0 else "bulk", name, cnt, gf, kmin, kmax)
    if d:
        line += "  | %-26s blocks %5d  %8.1f us" % d
        if kind == 1 and d[2] > 0:
            line += "  %5.1f TF" % (gf / d[2] * 1e3)
            tot[s][0] += gf; tot[s][1] += d[2]
    print(line)
for s in (0, 1):
    if tot[s][1] > 0:
        print("%s gemm total: %.1f Gflop in %.1f us = %.1f TFLOP/s (executed flops incl. masked triangles)" % ("main" if s == 0 else "bulk", tot[s][0], tot[s][1], tot[s][0] / tot[s][1] * 1e3))
