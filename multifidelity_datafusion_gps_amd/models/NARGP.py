import numpy as np

from ..MFDataFusion import MultifidelityDataFusion


class NARGP(MultifidelityDataFusion):
    """Nonlinear autoregressive multi-fidelity GP: high-fidelity inputs are augmented with the low-fidelity
    prediction only (no derivative stencil: n = 0, tau = 0) and the composite kernel k1*k2 + k3 is used
    (preset of /root/reference/src/models/NARGP.py:15-21)."""

    def __init__(self, input_dim: int, f_exact: callable, f_low: callable, name: str = 'NARGP',
                 lower_bound: np.ndarray = None, upper_bound: np.ndarray = None, lf_X: np.ndarray = None,
                 lf_Y: np.ndarray = None, lf_hf_adapt_ratio: int = 1, eps: float = 1e-8, add_noise: bool = False,
                 **kwargs):
        super().__init__(name=name, input_dim=input_dim, num_derivatives=0, tau=0, f_exact=f_exact,
                         lower_bound=lower_bound, upper_bound=upper_bound, f_low=f_low, lf_X=lf_X, lf_Y=lf_Y,
                         lf_hf_adapt_ratio=lf_hf_adapt_ratio, use_composite_kernel=True, eps=eps,
                         add_noise=add_noise, **kwargs)
