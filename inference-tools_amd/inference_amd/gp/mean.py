"""
Mean-function plugins of the GP path — counterpart of `inference/gp/mean.py`
(reference).  A mean function maps hyper-parameters to the prior mean vector of the
training points; it is O(N d) work and stays on the host: `GpRegressor` hands the
resulting vector to the device as the `mu` argument of gpmi_fit / gpmi_lml
(include/gpmi.h).

All three concrete classes are linear in their parameters, mu = Phi(x) . theta, so
they share one implementation built around the feature matrix Phi:

    ConstantMean   Phi = [1]                               (mean.py:31-51)
    LinearMean     Phi = [1, x - <x>]                      (mean.py:54-83)
    QuadraticMean  Phi = [1, x - <x>, (x - <x>)^2]         (mean.py:86-126)

Public names, constructor arguments, `bounds` / `n_params` / `hyperpar_labels`
attributes and method signatures are those of the reference.
"""
from abc import ABC, abstractmethod

import numpy as np
from numpy import ndarray


class MeanFunction(ABC):
    """The plugin contract `GpRegressor` relies on (mean.py:5-28)."""

    bounds = None
    n_params: int
    hyperpar_labels: list

    @abstractmethod
    def pass_spatial_data(self, x: ndarray):
        """Receive the (N, d) training coordinates."""

    @abstractmethod
    def estimate_hyperpar_bounds(self, y: ndarray):
        """Fill `self.bounds` from the training values."""

    @abstractmethod
    def __call__(self, q, theta: ndarray):
        """Prior mean at a single point q."""

    @abstractmethod
    def build_mean(self, theta: ndarray):
        """Prior mean vector at the training points."""

    @abstractmethod
    def mean_and_gradients(self, theta: ndarray):
        """(mean vector, list of d mean / d theta_j vectors)."""


class _FeatureMean(MeanFunction):
    """mu = Phi . theta with Phi made of an intercept and `degree` powers of the centred coordinates."""

    degree = 0

    def __init__(self, hyperpar_bounds=None):
        self.bounds = hyperpar_bounds
        if self.degree == 0:  # independent of the data: known before pass_spatial_data
            self.n_params = 1
            self.hyperpar_labels = self._labels(0)

    # -- feature map ----------------------------------------------------------------
    def _features(self, centred: ndarray) -> ndarray:
        cols = [np.ones((centred.shape[0], 1))]
        cols.extend(centred**p for p in range(1, self.degree + 1))
        return np.hstack(cols)

    def pass_spatial_data(self, x: ndarray):
        self.n_data, d = x.shape
        self.x_mean = x.mean(axis=0)
        centred = x - self.x_mean[None, :]
        self._phi = self._features(centred)
        self._span = centred.max(axis=0) - centred.min(axis=0)
        self.n_params = 1 + self.degree * d
        self.hyperpar_labels = self._labels(d)
        # attribute names of the reference's classes
        if self.degree >= 1:
            self.dx = centred
        if self.degree >= 2:
            self.dx_sqr = centred**2
            self.lin_slc = slice(1, d + 1)
            self.quad_slc = slice(d + 1, 2 * d + 1)

    def estimate_hyperpar_bounds(self, y: ndarray):
        lo, hi = y.min(), y.max()
        w = hi - lo
        if self.degree == 0:
            self.bounds = [(lo - w, hi + w)]
            return
        slope = 10 * w / self._span
        pairs = [(-b, b) for b in slope]
        self.bounds = [(lo - 2 * w, hi + 2 * w)] + pairs * self.degree

    # -- evaluation --------------------------------------------------------------------
    def __call__(self, q, theta: ndarray):
        if self.degree == 0:
            return theta[0]
        dq = np.atleast_2d(q - self.x_mean)
        return (self._features(dq) @ theta).squeeze()

    def at_points(self, points: ndarray, theta: ndarray) -> ndarray:
        """Prior mean at every row of `points` (M, d) in one shot — what `__call__` gives point by point."""
        if self.degree == 0:
            return np.full(points.shape[0], float(theta[0]))
        return self._features(points - self.x_mean[None, :]) @ np.asarray(theta, dtype=float)

    def build_mean(self, theta: ndarray):
        if self.degree == 0:
            return np.zeros(self.n_data) + theta[0]
        return self._phi @ np.asarray(theta, dtype=float)

    def mean_and_gradients(self, theta: ndarray):
        return self.build_mean(theta), list(self._phi.T)


class ConstantMean(_FeatureMean):
    degree = 0

    def _labels(self, d):
        return ["ConstantMean"]


class LinearMean(_FeatureMean):
    degree = 1

    def _labels(self, d):
        return ["LinearMean background"] + [f"LinearMean gradient {i}" for i in range(d)]


class QuadraticMean(_FeatureMean):
    degree = 2

    def _labels(self, d):
        return (
            ["mean_background"]
            + [f"mean_linear_coeff_{i}" for i in range(d)]
            + [f"mean_quadratic_coeff_{i}" for i in range(d)]
        )
