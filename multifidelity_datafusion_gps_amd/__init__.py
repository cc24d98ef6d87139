"""MI355X-native multi-fidelity GP path behind the AbstractMFGP / MultifidelityDataFusion surface.

(The distribution is named `multifidelity-datafusion-gps_amd`; a hyphen is not a legal Python
identifier, so the importable package is `multifidelity_datafusion_gps_amd` and the hyphenated path is a
symlink to it.)
"""
from .abstractMFGP import AbstractMFGP
from .MFDataFusion import MultifidelityDataFusion
from .models import GPDF, GPDFC, NARGP
from . import adaptation_maximizers, engine, gpc, sharding
from .adaptation_maximizers import AbstractMaximizer, DIRECT1Maximizer, PanelMaximizer, ScipyDirectMaximizer
from .augm_iterators import AbstractAugmIterator, BackwardAugmentation, EvenAugmentation

__all__ = ["AbstractMFGP", "MultifidelityDataFusion", "NARGP", "GPDF", "GPDFC", "engine", "gpc", "sharding",
           "AbstractMaximizer", "DIRECT1Maximizer", "PanelMaximizer", "ScipyDirectMaximizer", "adaptation_maximizers",
           "AbstractAugmIterator",
           "BackwardAugmentation", "EvenAugmentation"]
