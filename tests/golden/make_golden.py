"""Generates tests/golden/*.npz from the CPU oracle (oracle/gp_oracle.py) on the seeded cases of
tests/cases.py.  The reference itself cannot be run here (GPy is not installed nor installable
offline), so these vectors certify self-consistency with GPy-1.9.9's documented
math, not GPy's output: parity is UNPINNED at the GPy boundary.  Re-run: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import gp_oracle as orc  # noqa: E402
import cases  # noqa: E402

for name in cases.GOLDEN_CASES:
    if os.path.exists(os.path.join(HERE, name + ".npz")) and "--all" not in sys.argv:
        continue          # existing vectors stay byte-identical in the history; --all regenerates everything
    c = cases.make_case(name)
    parts, theta, noise = c["parts"], np.array(c["theta"], float), float(c["noise"])
    st = orc.inference(parts, theta, noise, c["X"], c["Y"])
    mu, var = orc.predict(parts, theta, noise, c["X"], st, c["Xs"])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), parts=np.array(parts), theta=theta, noise=noise,
                        X=c["X"], Y=c["Y"], Xs=c["Xs"], K=st["K"], L=st["L"], alpha=st["alpha"],
                        logdet=st["logdet"], nlml=st["nlml"], grad=st["grad"], mean=mu, var=var)
    print(name, "nlml=%.12g" % st["nlml"], "cond~%.3g" % np.linalg.cond(st["K"] + (noise + 1e-8) * np.eye(len(c["Y"]))))

# mid-size vectors: N = 512 .. 1024 (4 .. 8 leaf blocks of the HIP factorisation), compact -- alpha, diag(L), NLML, gradient,
# predictive mean and BOTH variance forms (GPy's explicit inverse = `var`, the triangular form = `var_stable`)
for name in cases.MID_GOLDEN_CASES:
    if os.path.exists(os.path.join(HERE, name + ".npz")) and "--all" not in sys.argv:
        continue
    c = cases.make_case(name)
    parts, theta, noise = c["parts"], np.array(c["theta"], float), float(c["noise"])
    st = orc.inference(parts, theta, noise, c["X"], c["Y"])
    mu, var = orc.predict(parts, theta, noise, c["X"], st, c["Xs"])
    _, var_s = orc.predict_stable(parts, theta, noise, c["X"], st, c["Xs"])
    ev = np.linalg.eigvalsh(st["K"] + (noise + 1e-8) * np.eye(len(c["Y"])))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), parts=np.array(parts), theta=theta, noise=noise,
                        X=c["X"], Y=c["Y"], Xs=c["Xs"], alpha=st["alpha"], diagL=np.diag(st["L"]).copy(),
                        logdet=st["logdet"], nlml=st["nlml"], grad=st["grad"], mean=mu, var=var, var_stable=var_s,
                        cond=ev[-1] / ev[0])
    print(name, "nlml=%.12g" % st["nlml"], "cond=%.3g" % (ev[-1] / ev[0]), "|var - var_stable| = %.2e" % np.abs(var - var_s).max())
