from .abstract_maximizer import AbstractMaximizer
from .direct import direct_minimize
from .DIRECT1_maximizer import DIRECT1Maximizer
from .scipydirect_wrapper import ScipyDirectMaximizer
