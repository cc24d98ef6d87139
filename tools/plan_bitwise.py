"""are the sweep plan's results independent of its blocking?  NLML + gradient under planner variants, compared bitwise.
usage: plan_bitwise.py N"""
import os
import subprocess
import sys

if len(sys.argv) > 2 and sys.argv[2] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    from multifidelity_datafusion_gps_amd._lib import Engine
    from tests import cases
    N = int(sys.argv[1])
    rng = np.random.default_rng(5)
    X = rng.uniform(size=(N, 4)); Xa = np.hstack([X, cases.lf_4d(X)[:, None]]); Y = cases.hf_4d(X)
    e = Engine(0); e.set_data(Xa, Y); e.set_kernel(cases.composite(4, 1))
    f, g = e.eval(np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.01 * Y.var())
    m, v = e.predict(Xa[:40])
    print(repr(float(f)), " ".join(repr(float(x)) for x in g), repr(float(m.sum())), repr(float(v.sum())))
    sys.exit(0)
N = sys.argv[1]
outs = {}
for env in ({}, {"MFGP_MACRO": "2"}, {"MFGP_MACRO": "4"}, {"MFGP_MACRO": "8", "MFGP_SHIFT": "0"}, {"MFGP_T128_MIN": "8"}, {"MFGP_T128_MIN": "100000"},
            {"MFGP_SHIFT": "0"}, {"MFGP_SHIFT": "1"}, {"MFGP_KINV_STREAM": "0"}, {"MFGP_CHAIN_SLIM": "0"}, {"MFGP_CHAIN_SLIM": "1"}, {"MFGP_PLAN": "levels"}):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), N, "child"], env=e, capture_output=True, text=True)
    outs[str(env)] = r.stdout.strip() or r.stderr[-300:]
ref = outs["{}"]
print("N =", N, "default:", ref[:60])
for k, v in outs.items():
    print("  %-50s %s" % (k, "BITWISE EQUAL" if v == ref else "differs: " + v[:80]))
