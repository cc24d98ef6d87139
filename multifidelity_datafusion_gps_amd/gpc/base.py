"""Polynomial-chaos post-processor interface.  A subclass provides the projection
(`calculate_coefficients`), the two moments and `update_order`; the base class offers the reference's convenience
calls `get_mean_var()` and `update_function(f)` (/root/reference/src/gpc/gpc_abstract.py)."""
from abc import ABC, abstractmethod


class AbstractGPC(ABC):
    """function: maps an (n, d) array of inputs to n (or (n, 1)) model outputs."""

    def __init__(self, function):
        self.function = function

    @abstractmethod
    def calculate_coefficients(self):
        """project `self.function` on the polynomial basis"""

    @abstractmethod
    def get_mean(self):
        """expectation of the expansion"""

    @abstractmethod
    def get_var(self):
        """variance of the expansion"""

    @abstractmethod
    def update_order(self, new_order):
        """change polynomial and quadrature order"""

    def get_mean_var(self):
        return (self.get_mean(), self.get_var())

    def update_function(self, function):
        """swap the model (e.g. after an adaptation round) and re-project"""
        self.function = function
        return self.calculate_coefficients()
