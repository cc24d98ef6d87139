import numpy as np

from ..MFDataFusion import MultifidelityDataFusion


class GPDFC(MultifidelityDataFusion):
    """GP with data fusion and the composite kernel k1*k2 + k3 (preset of
    /root/reference/src/models/GPDFC.py:16-22)."""

    def __init__(self, input_dim: int, tau: float, num_derivatives: int, f_exact: callable, f_low: callable,
                 name: str = 'GPDFC', lower_bound: np.ndarray = None, upper_bound: np.ndarray = None,
                 lf_X: np.ndarray = None, lf_Y: np.ndarray = None, lf_hf_adapt_ratio: int = 1, eps: float = 1e-8,
                 add_noise: bool = False, **kwargs):
        super().__init__(name=name, input_dim=input_dim, num_derivatives=num_derivatives, tau=tau, f_exact=f_exact,
                         lower_bound=lower_bound, upper_bound=upper_bound, f_low=f_low, lf_X=lf_X, lf_Y=lf_Y,
                         lf_hf_adapt_ratio=lf_hf_adapt_ratio, use_composite_kernel=True, eps=eps,
                         add_noise=add_noise, **kwargs)

    def lengthscale_hyperparams(self):
        """(l1, l2, l3) read through kernel.to_dict() exactly as the reference's plot helper does
        (src/models/GPDFC.py:26-29): l1 = k3 (additive part), l2 = k1 (augmentation), l3 = k2 (inputs)."""
        kern = self.kernel.to_dict()
        lengthscale1 = kern.get("parts").get(1).get("lengthscale")[0]
        lengthscale2 = kern.get("parts").get(0).get("parts").get(0).get("lengthscale")[0]
        lengthscale3 = kern.get("parts").get(0).get("parts").get(1).get("lengthscale")[0]
        return lengthscale1, lengthscale2, lengthscale3
