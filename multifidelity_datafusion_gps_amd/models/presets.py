"""The three constructor presets of the reference (`src/models/{NARGP,GPDF,GPDFC}.py`), as thin subclasses of
MultifidelityDataFusion that only pin (num_derivatives, tau, use_composite_kernel):

    NARGP : n = 0, tau = 0, composite kernel k1*k2 + k3   -> X_aug = [X | f_low(X)]          (NARGP.py:15-21)
    GPDF  : n, tau free,  ONE isotropic RBF over d+c cols  -> derivative stencil as inputs    (GPDF.py:15-21)
    GPDFC : n, tau free,  composite kernel                 -> both                            (GPDFC.py:16-22)

Positional argument order follows the reference so existing call sites keep working:
    NARGP(input_dim, f_exact, f_low, ...)      GPDF / GPDFC(input_dim, tau, num_derivatives, f_exact, f_low, ...)
Extra keyword arguments (seed, comm, engines, adapt_maximizer, ...) go through to MultifidelityDataFusion.
"""
from ..MFDataFusion import MultifidelityDataFusion



def _forward(kwargs, name):
    out = {"name": name, "lf_hf_adapt_ratio": 1, "eps": 1e-8, "add_noise": False}
    out.update(kwargs)
    return out


class NARGP(MultifidelityDataFusion):
    """Nonlinear autoregressive multi-fidelity GP: inputs augmented with the low-fidelity prediction only."""

    def __init__(self, input_dim, f_exact, f_low, name='NARGP', **kwargs):
        MultifidelityDataFusion.__init__(self, input_dim=input_dim, num_derivatives=0, tau=0, f_exact=f_exact,
                                         f_low=f_low, use_composite_kernel=True, **_forward(kwargs, name))


class GPDF(MultifidelityDataFusion):
    """GP with data fusion: low-fidelity values on a backward stencil (n steps of size tau) as extra inputs, single RBF."""

    def __init__(self, input_dim, tau, num_derivatives, f_exact, f_low, name='GPDF', **kwargs):
        MultifidelityDataFusion.__init__(self, input_dim=input_dim, num_derivatives=num_derivatives, tau=tau,
                                         f_exact=f_exact, f_low=f_low, use_composite_kernel=False,
                                         **_forward(kwargs, name))


class GPDFC(MultifidelityDataFusion):
    """GPDF with the composite NARGP kernel."""

    def __init__(self, input_dim, tau, num_derivatives, f_exact, f_low, name='GPDFC', **kwargs):
        MultifidelityDataFusion.__init__(self, input_dim=input_dim, num_derivatives=num_derivatives, tau=tau,
                                         f_exact=f_exact, f_low=f_low, use_composite_kernel=True,
                                         **_forward(kwargs, name))

    def lengthscale_hyperparams(self):
        """(l1, l2, l3) through kernel.to_dict(), the access path of the reference's plot helper
        (src/models/GPDFC.py:26-29): l1 = additive part k3, l2 = augmentation part k1, l3 = input part k2."""
        tree = self.kernel.to_dict()["parts"]
        product = tree[0]["parts"]
        return tree[1]["lengthscale"][0], product[0]["lengthscale"][0], product[1]["lengthscale"][0]
