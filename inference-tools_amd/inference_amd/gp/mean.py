"""
Mean-function classes — host-side mirror of `inference/gp/mean.py` (reference).
Mean vectors are O(N d) and are evaluated on the host, then handed to the device
as the `mu` argument of gpmi_fit / gpmi_lml (include/gpmi.h).
"""
from abc import ABC, abstractmethod

import numpy as np
from numpy import ndarray


class MeanFunction(ABC):
    """Plugin contract of the reference (mean.py:5-28)."""

    bounds = None
    n_params: int
    hyperpar_labels: list

    @abstractmethod
    def pass_spatial_data(self, x: ndarray):
        pass

    @abstractmethod
    def estimate_hyperpar_bounds(self, y: ndarray):
        pass

    @abstractmethod
    def __call__(self, q, theta: ndarray):
        pass

    @abstractmethod
    def build_mean(self, theta: ndarray):
        pass

    @abstractmethod
    def mean_and_gradients(self, theta: ndarray):
        pass


class ConstantMean(MeanFunction):
    """mu(x) = theta_0 (mean.py:31-51)."""

    def __init__(self, hyperpar_bounds=None):
        self.bounds = hyperpar_bounds
        self.n_params = 1
        self.hyperpar_labels = ["ConstantMean"]

    def pass_spatial_data(self, x: ndarray):
        self.n_data = x.shape[0]

    def estimate_hyperpar_bounds(self, y: ndarray):
        lo, hi = y.min(), y.max()
        w = hi - lo
        self.bounds = [(lo - w, hi + w)]

    def __call__(self, q, theta: ndarray):
        return theta[0]

    def build_mean(self, theta: ndarray):
        return np.zeros(self.n_data) + theta[0]

    def mean_and_gradients(self, theta: ndarray):
        return self.build_mean(theta), [np.ones(self.n_data)]


class LinearMean(MeanFunction):
    """mu(x) = theta_0 + (x - <x>) . theta_1: (mean.py:54-83)."""

    def __init__(self, hyperpar_bounds=None):
        self.bounds = hyperpar_bounds

    def pass_spatial_data(self, x: ndarray):
        self.x_mean = x.mean(axis=0)
        self.dx = x - self.x_mean[None, :]
        self.n_data, d = x.shape
        self.n_params = 1 + d
        self.hyperpar_labels = ["LinearMean background"] + [
            f"LinearMean gradient {i}" for i in range(d)
        ]

    def estimate_hyperpar_bounds(self, y: ndarray):
        w = y.max() - y.min()
        slope = 10 * w / (self.dx.max(axis=0) - self.dx.min(axis=0))
        self.bounds = [(y.min() - 2 * w, y.max() + 2 * w)] + [(-b, b) for b in slope]

    def __call__(self, q, theta: ndarray):
        return theta[0] + np.dot(q - self.x_mean, theta[1:]).squeeze()

    def build_mean(self, theta: ndarray):
        return theta[0] + np.dot(self.dx, theta[1:])

    def mean_and_gradients(self, theta: ndarray):
        return self.build_mean(theta), [np.ones(self.n_data), *self.dx.T]


class QuadraticMean(MeanFunction):
    """mu(x) = theta_0 + dx . lin + dx^2 . quad (mean.py:86-126)."""

    def __init__(self, hyperpar_bounds=None):
        self.bounds = hyperpar_bounds

    def pass_spatial_data(self, x: ndarray):
        self.n_data, d = x.shape
        self.x_mean = x.mean(axis=0)
        self.dx = x - self.x_mean[None, :]
        self.dx_sqr = self.dx**2
        self.n_params = 1 + 2 * d
        self.hyperpar_labels = (
            ["mean_background"]
            + [f"mean_linear_coeff_{i}" for i in range(d)]
            + [f"mean_quadratic_coeff_{i}" for i in range(d)]
        )
        self.lin_slc = slice(1, d + 1)
        self.quad_slc = slice(d + 1, 2 * d + 1)

    def estimate_hyperpar_bounds(self, y: ndarray):
        w = y.max() - y.min()
        slope = 10 * w / (self.dx.max(axis=0) - self.dx.min(axis=0))
        pairs = [(-b, b) for b in slope]
        self.bounds = [(y.min() - 2 * w, y.max() + 2 * w)] + pairs + pairs

    def __call__(self, q, theta: ndarray):
        dq = q - self.x_mean
        lin = np.dot(dq, theta[self.lin_slc]).squeeze()
        quad = np.dot(dq**2, theta[self.quad_slc]).squeeze()
        return theta[0] + lin + quad

    def build_mean(self, theta: ndarray):
        return theta[0] + np.dot(self.dx, theta[self.lin_slc]) + np.dot(self.dx_sqr, theta[self.quad_slc])

    def mean_and_gradients(self, theta: ndarray):
        return self.build_mean(theta), [np.ones(self.n_data), *self.dx.T, *self.dx_sqr.T]
