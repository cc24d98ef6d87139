"""The model constructions of the reference's own experiment scripts (/root/reference/tests/utils.py:30-47 `create_data`,
`create_mfgp_obj`; functions /root/reference/tests/test_mfgp_adapt_{2,4}d.py:9-25) restated as data-only problem
definitions shared by the fixture generator (which builds them with the REFERENCE's classes) and the replay tests (which
build them with this package's classes)."""
import numpy as np

from tests import cases


def hf_2d_T(x):
    return np.atleast_2d(cases.hf_2d(np.atleast_2d(x))).T


def lf_2d_T(x):
    return np.atleast_2d(cases.lf_2d(np.atleast_2d(x))).T


def hf_4d_T(x):
    return np.atleast_2d(cases.hf_4d(np.atleast_2d(x))).T


def lf_4d_T(x):
    return np.atleast_2d(cases.lf_4d(np.atleast_2d(x))).T


# name -> (model class name, constructor kwargs factory, dim, n_hf, n_test, global numpy seed)
PROBLEMS = {
    # tests/utils.py:38-47 with the 2-D functions: models.GPDF(dim, 0.001, 2, hf, lf, add_noise=True) etc.
    "gpdf_2d": dict(cls="GPDF", args=lambda: (2, 0.001, 2, hf_2d_T, lf_2d_T), kw=dict(add_noise=True), dim=2, n_hf=12, n_test=25, seed=10),
    "nargp_2d": dict(cls="NARGP", args=lambda: (2, hf_2d_T, lf_2d_T), kw=dict(add_noise=True), dim=2, n_hf=12, n_test=25, seed=11),
    "gpdfc_2d": dict(cls="GPDFC", args=lambda: (2, 0.001, 2, hf_2d_T, lf_2d_T), kw=dict(add_noise=True), dim=2, n_hf=12, n_test=25, seed=12),
    "nargp_4d": dict(cls="NARGP", args=lambda: (4, hf_4d_T, lf_4d_T), kw=dict(add_noise=False), dim=4, n_hf=24, n_test=30, seed=13),
    # a data-driven low-fidelity level (src/abstractMFGP.py:95-104): lf_X / lf_Y instead of a function
    "nargp_2d_datalf": dict(cls="NARGP", args=lambda: (2, hf_2d_T, None), kw=dict(add_noise=False), dim=2, n_hf=10, n_test=20, seed=14,
                            n_lf=30),
}


def make_inputs(name):
    """create_data of the reference (tests/utils.py:30-36): uniform draws from the GLOBAL numpy generator, in its order"""
    p = PROBLEMS[name]
    np.random.seed(p["seed"])
    dim = p["dim"]
    X_lf = np.random.uniform(low=0, high=1, size=(p.get("n_lf", 100), dim))
    X_hf = np.random.uniform(low=0, high=1, size=(p["n_hf"], dim))
    X_test = np.random.uniform(low=0, high=1, size=(p["n_test"], dim))
    return X_lf, X_hf, X_test


def build_and_run(name, models, recorder, adapt_steps=0, model_kw=None):
    """construct -> fit -> predict (-> adapt(adapt_steps) -> predict) with the classes of `models` (the reference's
    src.models or this package's models); returns a dict of plain results.  The restart draws come from the global numpy
    generator, re-seeded here so that both host layers consume the same stream."""
    p = PROBLEMS[name]
    X_lf, X_hf, X_test = make_inputs(name)
    kw = dict(p["kw"])
    if "n_lf" in p:
        kw.update(lf_X=X_lf, lf_Y=lf_2d_T(X_lf))
    kw.update(model_kw or {})
    np.random.seed(p["seed"] + 1000)
    model = getattr(models, p["cls"])(*p["args"](), **kw)
    model.fit(X_hf)
    mean, var = model.predict(X_test)
    out = dict(X_hf=X_hf, X_test=X_test, mean=np.asarray(mean), var=np.asarray(var),
               theta=np.array([float(q) for q in _param_values(model.hf_model)]))
    if adapt_steps:
        recorder.quiet = True
        model.adapt(adapt_steps)
        recorder.quiet = False
        m2, v2 = model.predict(X_test)
        out.update(adapt_hf_X=np.asarray(model.hf_X), adapt_mean=np.asarray(m2), adapt_var=np.asarray(v2),
                   adapt_n_predict=recorder.n_predict, adapt_predict_sha=recorder.predict_hash.hexdigest())
    return out


def _param_values(gp_model):
    return [p.value for p in gp_model.parameters()]


def run_gpc_rounds(name, models, recorder, driver_cls, gpc_cls, num_adapts=2, model_kw=None):
    """The reference's experiment scripts in miniature (tests/utils.py:75-86 of the reference): fit, wrap the model's posterior mean in a
    polynomial-chaos object, and let the round-based driver (`driver_cls`: the REFERENCE's src/gpc/mfgp_gpc.py::MFGP_GPC in the fixture
    generator, this package's gpc.MFGP_GPC in the replay test) alternate adaptation and moment refresh.  The polynomial-chaos object is
    this package's LegendreGPC in both runs (the reference's is a chaospy wrapper; chaospy is not available) -- what the fixture pins
    is the DRIVER: the order of adapt / update_function / get_mean / get_var / get_mse calls and the cost bookkeeping.  -> histories."""
    p = PROBLEMS[name]
    X_lf, X_hf, X_test = make_inputs(name)
    kw = dict(p["kw"])
    kw.update(model_kw or {})
    np.random.seed(p["seed"] + 2000)
    model = getattr(models, p["cls"])(*p["args"](), **kw)
    model.fit(X_hf)
    hf = {2: hf_2d_T, 4: hf_4d_T}[p["dim"]]
    pce = gpc_cls(lambda x: model.predict(x)[0], np.zeros(p["dim"]), np.ones(p["dim"]), polynomial_order=4, quadrature_order=4)
    recorder.quiet = True
    drv = driver_cls(model, pce, num_adapts, float(len(X_hf)), X_test=X_test, Y_test=hf(X_test))
    drv.adapt()
    recorder.quiet = False
    return dict(mean_history=np.array(drv.mean_history, dtype=np.float64), var_history=np.array(drv.var_history, dtype=np.float64),
                cost_history=np.array(drv.cost_history, dtype=np.float64), mse_history=np.array(drv.mse_history, dtype=np.float64),
                hf_X=np.asarray(model.hf_X), n_predict=recorder.n_predict, predict_sha=recorder.predict_hash.hexdigest())
