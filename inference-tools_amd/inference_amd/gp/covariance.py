"""
Covariance-function classes of the GP path — host-side mirror of
`inference/gp/covariance.py` (reference): same class names, constructor
arguments, hyper-parameter layout (natural logs), labels, bounds estimates and
method names, so that `kernel=SquaredExponential`, `RationalQuadratic() +
WhiteNoise()` etc. are drop-in.

Differences in mechanism (not in results):
  * `pass_spatial_data` keeps only x (N x d).  The reference's N x N x d tensors
    `dx` / `distances` (covariance.py:218-219, 315-316) are never formed; the
    covariance matrix is produced on the device by the tiled HIP kernel of
    `csrc/kbuild.hip` through `gpmi_*` (include/gpmi.h).
  * `estimate_hyperpar_bounds` obtains mean|dx_k| over all ordered pairs from a
    sort (O(N log N)) instead of reducing an N x N array.
`GpRegressor` recognises these classes and drives the device directly; the
plugin methods (`__call__`, `build_covariance`, `covariance_and_gradients`) are
kept for code that calls them and evaluate on the device as well.
"""
from abc import ABC, abstractmethod

import numpy as np
from numpy import exp, log, ndarray

from inference_amd import _lib


class CovarianceFunction(ABC):
    """Plugin contract of the reference (covariance.py:8-44)."""

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
    def __call__(self, u: ndarray, v: ndarray, theta: ndarray) -> ndarray:
        pass

    @abstractmethod
    def build_covariance(self, theta: ndarray) -> ndarray:
        pass

    @abstractmethod
    def covariance_and_gradients(self, theta: ndarray):
        pass

    def __add__(self, other):
        mine = self.components if isinstance(self, CompositeCovariance) else [self]
        theirs = other.components if isinstance(other, CompositeCovariance) else [other]
        return CompositeCovariance([*mine, *theirs])

    def gradient_terms(self, v, x, theta):
        raise NotImplementedError(
            f"""
            Gradient calculations are not yet available for the
            {type(self)} covariance function.
            """
        )

    def get_bounds(self):
        return self.bounds


def _pairwise_abs_mean_and_range(col: ndarray):
    """mean over all N^2 ordered pairs of |x_i - x_j| (zeros on the diagonal
    included) and max(x_i - x_j) = range — what covariance.py:236-237 reduces
    from the N x N array — via the sorted-order identity
    sum_{i,j} |x_i - x_j| = 2 * sum_i (2 i - N + 1) x_(i)."""
    s = np.sort(col)
    n = s.size
    total = 2.0 * np.dot(2.0 * np.arange(n) - n + 1.0, s)
    return total / (float(n) * float(n)), s[-1] - s[0]


class _StationaryDeviceKernel(CovarianceFunction):
    """Shared machinery of the two device-resident stationary kernels."""

    _gpmi_kernel = None  # GPMI_KERNEL_* id
    _n_shape_params = 0  # parameters between the amplitude and the length-scales

    def __init__(self, hyperpar_bounds=None):
        self.bounds = hyperpar_bounds
        self.x = None
        self._engine = None

    def pass_spatial_data(self, x: ndarray):
        self.x = np.ascontiguousarray(x, dtype=float)
        self._engine = None
        d = self.x.shape[1]
        self.n_params = d + 1 + self._n_shape_params
        self.hyperpar_labels = self._labels(d)

    def _scale_bounds(self):
        out = []
        for i in range(self.x.shape[1]):
            mean_abs, rng = _pairwise_abs_mean_and_range(self.x[:, i])
            out.append((log(mean_abs) - 4, log(rng) + 2))
        return out

    # -- device evaluation of the plugin methods --------------------------------------
    def _own_engine(self):
        if self._engine is None:
            from inference_amd._engine import GpEngine

            self._engine = GpEngine(self.x, np.zeros(self.x.shape[0]))
        return self._engine

    def __call__(self, u: ndarray, v: ndarray, theta: ndarray) -> ndarray:
        from inference_amd._engine import GpEngine

        u = np.ascontiguousarray(u, dtype=float)
        v = np.ascontiguousarray(v, dtype=float)
        eng = GpEngine(v, np.zeros(v.shape[0]))
        try:
            return eng.cross_covariance(self._gpmi_kernel, theta, u)
        finally:
            eng.close()

    def build_covariance(self, theta: ndarray) -> ndarray:
        return self._own_engine().covariance(self._gpmi_kernel, theta)


class SquaredExponential(_StationaryDeviceKernel):
    r"""
    Squared-exponential covariance (reference: covariance.py:181-279)

       K(u, v) = A^2 exp( -1/2 sum_i ((u_i - v_i) / l_i)^2 ),   theta = [ln A, ln l_1 .. ln l_n]

    :param hyperpar_bounds: optional list of (lower, upper) tuples, one per parameter;
        estimated from the data when omitted.
    """

    _gpmi_kernel = _lib.KERNEL_SE
    _n_shape_params = 0

    def _labels(self, d):
        return ["SqrExp log-amplitude"] + [f"SqrExp log-scale {i}" for i in range(d)]

    def estimate_hyperpar_bounds(self, y: ndarray):
        s = log(y.std())
        self.bounds = [(s - 4, s + 4)] + self._scale_bounds()

    def gradient_terms(self, v: ndarray, x: ndarray, theta: ndarray):
        """(A, R) of the predictive-gradient expressions (covariance.py:257-266)."""
        a = exp(theta[0])
        scales = exp(theta[1:])
        A = (x - v[None, :]) / scales[None, :] ** 2
        return A.T, (a / scales) ** 2

    def covariance_and_gradients(self, theta: ndarray):
        """K and dK/dtheta_j as dense matrices (covariance.py:268-276).  The regressor never
        calls this (its LML gradient contracts dK on the fly on the device); it is
        provided for plugin users and is O(N^2 d) in memory like the reference."""
        K = self.build_covariance(theta)
        scales = exp(theta[1:])
        grads = [2.0 * K]
        for i, l in enumerate(scales):
            dx = self.x[:, None, i] - self.x[None, :, i]
            grads.append((dx**2 / l**2) * K)
        return K, grads


class RationalQuadratic(_StationaryDeviceKernel):
    r"""
    Rational-quadratic covariance (reference: covariance.py:282-368)

       K(u, v) = A^2 (1 + 1/(2 alpha) sum_i ((u_i - v_i)/l_i)^2)^(-alpha),
       theta = [ln A, ln alpha, ln l_1 .. ln l_n]
    """

    _gpmi_kernel = _lib.KERNEL_RQ
    _n_shape_params = 1

    def _labels(self, d):
        return ["RQ log-amplitude", "RQ log-alpha"] + [f"RQ log-scale {i}" for i in range(d)]

    def estimate_hyperpar_bounds(self, y: ndarray):
        s = log(y.std())
        self.bounds = [(s - 4, s + 4), (-2, 6)] + self._scale_bounds()

    def covariance_and_gradients(self, theta: ndarray):
        """Dense K and gradients (covariance.py:350-365); see SquaredExponential's note."""
        K = self.build_covariance(theta)
        q = exp(theta[1])
        scales = exp(theta[2:])
        Z = np.zeros_like(K)
        half_sq = []
        for i, l in enumerate(scales):
            dx = self.x[:, None, i] - self.x[None, :, i]
            half_sq.append(0.5 * dx**2 / l**2)
            Z += half_sq[-1]
        F = 1 + Z / q
        grads = [2.0 * K, -K * (log(F) * q - Z / F)]
        G = 2 * K / F
        grads.extend(G * h for h in half_sq)
        return K, grads


class WhiteNoise(CovarianceFunction):
    r"""
    Independent Gaussian noise, K = delta_ij sigma_n^2 with theta = [ln sigma_n]
    (reference: covariance.py:108-178).  Used as `SquaredExponential() + WhiteNoise()`;
    on the device it is a diagonal add fused into the covariance build.
    """

    def __init__(self, hyperpar_bounds=None):
        self.bounds = hyperpar_bounds
        self.n_params = 1
        self.hyperpar_labels = ["WhiteNoise log-sigma"]
        self._n = 0

    def pass_spatial_data(self, x: ndarray):
        self._n = x.shape[0]

    def estimate_hyperpar_bounds(self, y: ndarray):
        s = log(np.ptp(y))
        self.bounds = [(s - 8, s + 2)]

    def __call__(self, u: ndarray, v: ndarray, theta):
        return np.zeros([u.shape[0], v.shape[0]])

    def build_covariance(self, theta):
        return exp(2 * theta[0]) * np.eye(self._n)

    def covariance_and_gradients(self, theta):
        K = self.build_covariance(theta)
        return K, [2.0 * K]


class HeteroscedasticNoise(CovarianceFunction):
    r"""
    Independent Gaussian noise with one standard deviation per data value,
    K = delta_ij sigma_i^2, theta = [ln sigma_1 .. ln sigma_N] (reference: covariance.py:608-690).
    Used as `SquaredExponential() + HeteroscedasticNoise()`.  On the device it is part of the diagonal term
    of the covariance build (the host adds exp(2 theta_i) to the data variances, `gpmi_set_noise`), and its
    N gradient components come from one vector, sigma_i^2 (alpha_i^2 - (K^-1)_ii) (`gpmi_lml_grad_qdiag`),
    instead of the reference's N dense N x N matrices.
    """

    def __init__(self, hyperpar_bounds=None):
        self.bounds = hyperpar_bounds
        self.n_params = 0
        self.hyperpar_labels = []

    def pass_spatial_data(self, x: ndarray):
        self.n_params = x.shape[0]
        self.hyperpar_labels = [f"log_sigma_{i + 1}" for i in range(self.n_params)]

    def estimate_hyperpar_bounds(self, y: ndarray):
        s = log(np.ptp(y))
        self.bounds = [(s - 8, s + 2)] * self.n_params

    def __call__(self, u: ndarray, v: ndarray, theta):
        # the reference sizes this block by u.size / v.size (covariance.py:671-672), which only works for
        # one spatial dimension; row counts give the same result there and the intended one for d > 1
        return np.zeros([np.atleast_2d(u).shape[0] if np.ndim(u) > 1 else np.size(u),
                         np.atleast_2d(v).shape[0] if np.ndim(v) > 1 else np.size(v)])

    def build_covariance(self, theta):
        return np.diag(exp(2 * np.asarray(theta)))

    def covariance_and_gradients(self, theta):
        # host fall-back only (N dense matrices, as the reference builds them); the device path never calls it
        var = exp(2 * np.asarray(theta))
        grads = []
        for i, v in enumerate(var):
            G = np.zeros([self.n_params, self.n_params])
            G[i, i] = 2.0 * v
            grads.append(G)
        return np.diag(var), grads


def slice_builder(lengths):
    out, lo = [], 0
    for n in lengths:
        out.append(slice(lo, lo + n))
        lo += n
    return out


class CompositeCovariance(CovarianceFunction):
    """Sum of covariance functions with concatenated parameters (covariance.py:47-105)."""

    def __init__(self, covariance_components):
        self.components = covariance_components
        self.bounds = None

    def pass_spatial_data(self, x: ndarray):
        for comp in self.components:
            comp.pass_spatial_data(x)
        self.slices = slice_builder([c.n_params for c in self.components])
        self.hyperpar_labels = [
            f"K{i + 1}: {s}" for i, comp in enumerate(self.components) for s in comp.hyperpar_labels
        ]
        self.n_params = sum(c.n_params for c in self.components)

    def estimate_hyperpar_bounds(self, y: ndarray):
        self.bounds = []
        for comp in self.components:
            if comp.bounds is None:
                comp.estimate_hyperpar_bounds(y)
            self.bounds.extend(comp.bounds)

    def __call__(self, u, v, theta):
        return sum(c(u, v, theta[s]) for c, s in zip(self.components, self.slices))

    def build_covariance(self, theta):
        return sum(c.build_covariance(theta[s]) for c, s in zip(self.components, self.slices))

    def covariance_and_gradients(self, theta):
        parts = [c.covariance_and_gradients(theta[s]) for c, s in zip(self.components, self.slices)]
        K = sum(p[0] for p in parts)
        grads = [g for p in parts for g in p[1]]
        return K, grads


def device_plan(cov):
    """How `GpRegressor` maps a covariance object onto the device kernels:
    returns (kernel_id, stationary_component, slice_of_its_theta, white_noise_index or None)
    or None when the object is not a supported combination: one of the stationary kernels, optionally plus
    one WhiteNoise and / or one HeteroscedasticNoise (see `heteroscedastic_slice`)."""
    if isinstance(cov, _StationaryDeviceKernel):
        return cov._gpmi_kernel, cov, slice(0, cov.n_params), None
    if isinstance(cov, CompositeCovariance):
        stat = [(i, c) for i, c in enumerate(cov.components) if isinstance(c, _StationaryDeviceKernel)]
        wn = [(i, c) for i, c in enumerate(cov.components) if isinstance(c, WhiteNoise)]
        het = [c for c in cov.components if isinstance(c, HeteroscedasticNoise)]
        if len(stat) == 1 and len(wn) <= 1 and len(het) <= 1 and len(stat) + len(wn) + len(het) == len(cov.components):
            i, c = stat[0]
            wn_index = cov.slices[wn[0][0]].start if wn else None
            return c._gpmi_kernel, c, cov.slices[i], wn_index
    return None


def heteroscedastic_slice(cov):
    """Slice of the HeteroscedasticNoise parameters inside the covariance parameter vector, or None."""
    if isinstance(cov, CompositeCovariance):
        for comp, sl in zip(cov.components, cov.slices):
            if isinstance(comp, HeteroscedasticNoise):
                return sl
    return None
