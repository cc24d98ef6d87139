import numpy as np

from ..MFDataFusion import MultifidelityDataFusion


class GPDF(MultifidelityDataFusion):
    """GP with data fusion: low-fidelity predictions plus an implicit-derivative stencil (n, tau) as extra
    inputs, one isotropic RBF over all columns (preset of /root/reference/src/models/GPDF.py:15-21)."""

    def __init__(self, input_dim: int, tau: float, num_derivatives: int, f_exact: callable, f_low: callable,
                 name: str = 'GPDF', lower_bound: np.ndarray = None, upper_bound: np.ndarray = None,
                 lf_X: np.ndarray = None, lf_Y: np.ndarray = None, lf_hf_adapt_ratio: int = 1, eps: float = 1e-8,
                 add_noise: bool = False, **kwargs):
        super().__init__(name=name, input_dim=input_dim, num_derivatives=num_derivatives, tau=tau, f_exact=f_exact,
                         lower_bound=lower_bound, upper_bound=upper_bound, f_low=f_low, lf_X=lf_X, lf_Y=lf_Y,
                         lf_hf_adapt_ratio=lf_hf_adapt_ratio, use_composite_kernel=False, eps=eps,
                         add_noise=add_noise, **kwargs)
