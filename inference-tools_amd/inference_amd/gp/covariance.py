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

from inspect import isclass

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
        if self._engine is not None:
            self._engine.close()
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

    def _cross_engine(self, v):
        """Device context holding the points `v` of the second argument of `__call__`: the training points (already
        uploaded: the `build_covariance` context) or the last other point set, kept until a different one arrives -
        a plugin user calling `cov(u, x, theta)` in a loop does not pay a context (streams, buffers) per call."""
        from inference_amd._engine import GpEngine

        if self.x is not None and v.shape == self.x.shape and np.array_equal(v, self.x):
            return self._own_engine()
        cached = getattr(self, "_cross", None)
        if cached is not None and cached[0].shape == v.shape and np.array_equal(cached[0], v):
            return cached[1]
        if cached is not None:
            cached[1].close()
        eng = GpEngine(v, np.zeros(v.shape[0]))
        self._cross = (v.copy(), eng)
        return eng

    def __call__(self, u: ndarray, v: ndarray, theta: ndarray) -> ndarray:
        u = np.ascontiguousarray(u, dtype=float)
        v = np.ascontiguousarray(v, dtype=float)
        return self._cross_engine(v).cross_covariance(self._gpmi_kernel, theta, u)

    def __getstate__(self):
        state = self.__dict__.copy()  # device contexts do not pickle: re-created on demand
        state["_engine"] = None
        state.pop("_cross", None)
        return state

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


class ChangePoint(CovarianceFunction):
    r"""
    Change-point covariance (reference: covariance.py:371-606): the input space is divided along one axis
    into regions, each with its own kernel, blended by logistic windows f_i(x) = 1 / (1 + exp(-(x - c_i) / w_i)):

       K(u, v) = sum_m g_m(u) g_m(v) K_m(u, v),   g_0 = 1 - f_0,  g_m = f_{m-1} (1 - f_m),  g_last = f_last

    theta = [theta_K0, theta_K1, .., c_0, w_0, c_1, w_1, ..].  Because the coefficients factorise over the two
    points, K is a sum of stationary kernels scaled by per-point weights: on the device every K_m comes from the
    ordinary covariance-build kernel and is folded in with its weights (`gpmi_fit_mix`, `gpmi_lml_mix`,
    `gpmi_lml_grad_mix`, `gpmi_predict_mix`); the weights cost O(N) and are formed here.

    :param kernels: the kernels of the regions, instances or classes (device path: SquaredExponential /
        RationalQuadratic, at most four).
    :param axis: the spatial axis along which the regions follow each other.
    :param location_bounds, width_bounds: optional (lower, upper) pairs, one per change-point.
    """

    def __init__(self, kernels, axis: int = 0, location_bounds=None, width_bounds=None):
        self.cov = [K() if isclass(K) and issubclass(K, CovarianceFunction) else K for K in kernels]
        for K in self.cov:
            if not isinstance(K, CovarianceFunction):
                raise TypeError(
                    "\n\n[ ChangePoint error ]\n>> Each of the specified covariance kernels must be an instance of"
                    "\n>> a class which inherits from the 'CovarianceFunction' abstract\n>> base-class.\n"
                )
        self.n_kernels = len(kernels)

        def pairs(bounds, what):
            if bounds is None:
                return None
            if len(bounds) != self.n_kernels - 1:
                raise ValueError(
                    f"\n\n[ ChangePoint error ]\n>> The length of '{what}' must be one less than the number of kernels\n"
                )
            for b in bounds:
                assert type(b) in [list, tuple, ndarray] and len(b) == 2 and b[1] > b[0]
            return list(bounds)

        self.location_bounds = pairs(location_bounds, "location_bounds")
        self.width_bounds = pairs(width_bounds, "width_bounds")
        self.axis = axis
        self.bounds = None

    def pass_spatial_data(self, x: ndarray):
        for K in self.cov:
            K.pass_spatial_data(x)
        counts = [K.n_params for K in self.cov] + [2] * (self.n_kernels - 1)
        self.n_params = sum(counts)
        slices = slice_builder(counts)
        self.cov_slc = slices[: self.n_kernels]
        self.cp_slc = slices[self.n_kernels:]
        self.hyperpar_labels = [f"ChngPnt K{i}: {lab}" for i, K in enumerate(self.cov) for lab in K.hyperpar_labels]
        for i in range(self.n_kernels - 1):
            self.hyperpar_labels += [f"ChngPnt{i} location", f"ChngPnt{i} width"]
        self.x_cp = x[:, self.axis]

    def estimate_hyperpar_bounds(self, y: ndarray):
        lo, hi = self.x_cp.min(), self.x_cp.max()
        span = hi - lo
        self.bounds = []
        for K in self.cov:
            K.estimate_hyperpar_bounds(y)
            self.bounds.extend(K.bounds)
        if self.location_bounds is None:
            self.location_bounds = [(lo, hi)] * (self.n_kernels - 1)
        if self.width_bounds is None:
            self.width_bounds = [(5e-3 * span, 0.5 * span)] * (self.n_kernels - 1)
        for loc, wid in zip(self.location_bounds, self.width_bounds):
            self.bounds += [loc, wid]

    # -- the per-point weights ---------------------------------------------------------------
    @staticmethod
    def logistic(x, theta):
        z = (x - theta[0]) / theta[1]
        return 1.0 / (1.0 + exp(-z))

    @staticmethod
    def logistic_and_gradient(x, theta):
        z = (x - theta[0]) / theta[1]
        f = 1.0 / (1.0 + exp(-z))
        dfdc = -f * (1 - f) / theta[1]
        return f, [dfdc, dfdc * z]

    def weights(self, axis_values, theta):
        """g_m at the given coordinates along the change-point axis: array (n_kernels, len(axis_values))."""
        g = [np.ones_like(axis_values, dtype=float)]
        for slc in self.cp_slc:
            f = self.logistic(axis_values, theta[slc])
            g[-1] = g[-1] * (1 - f)
            g.append(f)
        return np.array(g)

    def device_terms(self, theta):
        """(kernel ids, sub-kernel parameter vectors) for the gpmi_*_mix entry points."""
        return [K._gpmi_kernel for K in self.cov], [np.asarray(theta[s], dtype=float) for s in self.cov_slc]

    # -- plugin methods (host composition of the sub-kernels' device results) --------------------
    def __call__(self, u: ndarray, v: ndarray, theta: ndarray) -> ndarray:
        gu, gv = self.weights(u[:, self.axis], theta), self.weights(v[:, self.axis], theta)
        return sum(K(u, v, theta[s]) * (a[:, None] * b[None, :]) for K, s, a, b in zip(self.cov, self.cov_slc, gu, gv))

    def build_covariance(self, theta: ndarray) -> ndarray:
        g = self.weights(self.x_cp, theta)
        return sum(K.build_covariance(theta[s]) * (a[:, None] * a[None, :]) for K, s, a in zip(self.cov, self.cov_slc, g))

    def covariance_and_gradients(self, theta: ndarray):
        parts = [K.covariance_and_gradients(theta[s]) for K, s in zip(self.cov, self.cov_slc)]
        coeffs, w_vals, w_grads = [1.0], [], []
        for slc in self.cp_slc:
            w, dw = self.logistic_and_gradient(self.x_cp, theta[slc])
            coeffs[-1] = coeffs[-1] * ((1 - w)[:, None] * (1 - w)[None, :])
            coeffs.append(w[:, None] * w[None, :])
            w_vals.append(w)
            w_grads.append(dw)
        K = sum(p[0] * c for p, c in zip(parts, coeffs))
        grads = [dK * c for p, c in zip(parts, coeffs) for dK in p[1]]
        for i, (w, dws) in enumerate(zip(w_vals, w_grads)):
            for dw in dws:  # covariance.py:588-593
                A = -dw[:, None] * (1 - w)[None, :]
                B = dw[:, None] * w[None, :]
                grads.append(parts[i][0] * (A + A.T) + parts[i + 1][0] * (B + B.T))
        return K, grads


def _is_device_mixture(comp):
    return (isinstance(comp, ChangePoint) and 2 <= comp.n_kernels <= 4
            and all(isinstance(K, _StationaryDeviceKernel) for K in comp.cov))


def device_plan(cov):
    """How `GpRegressor` maps a covariance object onto the device kernels:
    returns (kernel_id, main_component, slice_of_its_theta, white_noise_index or None)
    or None when the object is not a supported combination: one stationary kernel (SquaredExponential /
    RationalQuadratic) or one ChangePoint over such kernels (kernel_id -1), optionally plus one WhiteNoise
    and / or one HeteroscedasticNoise (see `heteroscedastic_slice`)."""
    if isinstance(cov, _StationaryDeviceKernel):
        return cov._gpmi_kernel, cov, slice(0, cov.n_params), None
    if _is_device_mixture(cov):
        return -1, cov, slice(0, cov.n_params), None
    if isinstance(cov, CompositeCovariance):
        main = [(i, c) for i, c in enumerate(cov.components)
                if isinstance(c, _StationaryDeviceKernel) or _is_device_mixture(c)]
        wn = [(i, c) for i, c in enumerate(cov.components) if isinstance(c, WhiteNoise)]
        het = [c for c in cov.components if isinstance(c, HeteroscedasticNoise)]
        if len(main) == 1 and len(wn) <= 1 and len(het) <= 1 and len(main) + len(wn) + len(het) == len(cov.components):
            i, c = main[0]
            wn_index = cov.slices[wn[0][0]].start if wn else None
            return (c._gpmi_kernel if isinstance(c, _StationaryDeviceKernel) else -1), c, cov.slices[i], wn_index
    return None


def heteroscedastic_slice(cov):
    """Slice of the HeteroscedasticNoise parameters inside the covariance parameter vector, or None."""
    if isinstance(cov, CompositeCovariance):
        for comp, sl in zip(cov.components, cov.slices):
            if isinstance(comp, HeteroscedasticNoise):
                return sl
    return None
