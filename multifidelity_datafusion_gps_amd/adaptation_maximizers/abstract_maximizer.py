"""Interface of the uncertainty maximisers (mirrors /root/reference/src/adaptation_maximizers/abstract_maximizer.py:5-28)."""
from abc import ABCMeta, abstractmethod

import numpy as np


class AbstractMaximizer(metaclass=ABCMeta):
    """maximize(model_predict, lower_bound, upper_bound) -> (x_opt, f_opt) where f_opt = -max variance
    (the reference minimises the negated predictive variance and returns that negated value)."""

    @abstractmethod
    def __init__(self):
        super().__init__()

    @abstractmethod
    def maximize(self, model_predict: callable, lower_bound: np.ndarray, upper_bound: np.ndarray):
        """model_predict maps (B, d) inputs to (means (B,1), variances (B,1))."""
