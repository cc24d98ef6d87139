"""Variance maximisers of the adaptation loop (`src.adaptation_maximizers` of the reference) over one batched DIRECT."""
from .abstract_maximizer import AbstractMaximizer  # noqa: F401
from .direct import direct_minimize, gablonsky_direct  # noqa: F401
from .DIRECT1_maximizer import DIRECT1Maximizer  # noqa: F401
from .panel_maximizer import PanelMaximizer  # noqa: F401
from .scipydirect_wrapper import ScipyDirectMaximizer  # noqa: F401

__all__ = ["AbstractMaximizer", "DIRECT1Maximizer", "PanelMaximizer", "ScipyDirectMaximizer", "direct_minimize", "gablonsky_direct"]
