from .base import AbstractGPC
from .legendre_gpc import LegendreGPC
from .mfgp_gpc import MFGP_GPC
