"""
`GpRegressor` — drop-in for `inference.gp.GpRegressor` (reference:
inference/gp/regression.py:16-612) whose covariance build, Cholesky
factorisation / solves and log-marginal-likelihood evaluations run on an MI355X
through the C-ABI of include/gpmi.h.

Same constructor signature, attribute names (`x, y, sig, cov, mean, hp_bounds,
hyperpars, K_xx, L, alpha, mu, ...`), return shapes, warnings and errors as the
reference.  Differences in mechanism only:
  * prediction is one batched device call instead of a Python loop over points
    (regression.py:208-214);
  * `K_xx`, `L` and `sig` are materialised lazily on first access (N^2 download);
  * hyper-parameter layout, the LML without the 2 pi constant, the -1e50
    sentinel, `sqrt(abs(var))` and the `.squeeze()`d gradient outputs are kept.
"""
from copy import copy
from inspect import isclass
from warnings import catch_warnings, simplefilter, warn

import numpy as np
from numpy import array, ndarray, sqrt, zeros
from numpy.linalg import LinAlgError
from numpy.random import random
from scipy.optimize import differential_evolution, fmin_l_bfgs_b

from inference_amd._engine import GpEngine
from inference_amd.gp import _messages as msg
from inference_amd.gp.covariance import heteroscedastic_slice, CovarianceFunction, SquaredExponential, device_plan
from inference_amd.gp.mean import ConstantMean, MeanFunction


class GpRegressor:
    """
    Gaussian-process regression in one or more dimensions (see the reference's
    docstring, regression.py:17-77, for the modelling description).

    :param x: (N, d) array (or array-like) of point coordinates; 1-D input means d = 1.
    :param y: (N,) array of values.
    :param y_err: (N,) standard deviations of the values (Gaussian errors).
    :param y_cov: (N, N) covariance of the values, alternative to ``y_err``.
    :param hyperpars: hyper-parameter vector ``[mean params | covariance params]``;
        when omitted it is found by maximising the marginal likelihood (or the
        leave-one-out likelihood if ``cross_val``).
    :param kernel: covariance class or instance (``SquaredExponential``,
        ``RationalQuadratic``, optionally ``+ WhiteNoise()``).
    :param mean: mean-function class or instance.
    :param bool cross_val: select hyper-parameters by LOO cross-validation.
    :param str optimizer: ``"bfgs"`` (multi-start L-BFGS-B) or ``"diffev"``.
    :param int n_processes: number of GPU workers for the multi-start search: inside a multi-GPU
        job (one process per GPU, `inference_amd.sharding`) the L-BFGS-B starts are sharded over the
        ranks; in a single process they run one after another on the device.
    :param int n_starts: number of L-BFGS-B starting positions.
    :param device: (extension) HIP device index; default ``LOCAL_RANK`` / 0.
    :param reserve: (extension) room for this many more training points on the device, see ``add_point``.
    """

    def __init__(
        self,
        x: ndarray,
        y: ndarray,
        y_err: ndarray = None,
        y_cov: ndarray = None,
        hyperpars: ndarray = None,
        kernel: CovarianceFunction = SquaredExponential,
        mean: MeanFunction = ConstantMean,
        cross_val: bool = False,
        optimizer: str = "bfgs",
        n_processes: int = 1,
        n_starts: int = None,
        device: int = None,
        reserve: int = 0,
        diffev_batched: bool = False,
    ):
        self.x, self.y = self._coerce_training_data(x, y)
        self._diffev_batched = bool(diffev_batched)
        self.n_points = self.y.size
        self.n_dimensions = self.x.shape[1]

        # data-error covariance: kept as a variance vector or the dense matrix the user gave
        self._noise_var, self._y_cov = self.check_error_data(y_err, y_cov)

        self.cov = kernel() if isclass(kernel) else kernel
        self.mean = mean() if isclass(mean) else mean

        self.cov.pass_spatial_data(self.x)
        self.mean.pass_spatial_data(self.x)
        if self.cov.bounds is None:
            self.cov.estimate_hyperpar_bounds(self.y)
        if self.mean.bounds is None:
            self.mean.estimate_hyperpar_bounds(self.y)
        self.hp_bounds = copy(self.mean.bounds)
        self.hp_bounds.extend(copy(self.cov.bounds))
        self.n_hyperpars = len(self.hp_bounds)
        self.mean_slice = slice(0, self.mean.n_params)
        self.cov_slice = slice(self.mean.n_params, self.n_hyperpars)
        self.hyperpar_labels = [*self.mean.hyperpar_labels, *self.cov.hyperpar_labels]

        plan = device_plan(self.cov)
        # A covariance object that only implements the plugin ABC (covariance.py:8-44): its own host methods
        # produce the dense matrices, the device does every O(N^3) step (gpmi_*_dense).  No CPU solve anywhere.
        self._generic = plan is None
        if self._generic:
            plan = (None, None, slice(0, self.cov.n_params), None)
        self._kernel_id, self._stat, self._stat_slice, self._wn_index = plan
        self._mix = self._stat if self._kernel_id == -1 else None  # ChangePoint: mixture entry points
        self._het_slice = None if self._generic else heteroscedastic_slice(self.cov)
        self._fit_noise = None
        self._device = device
        self._reserve = int(reserve)
        self._engine = None
        self._K_cache = None
        self._L_cache = None

        if cross_val:
            self.model_selector = self.loo_likelihood
            self.model_selector_gradient = self.loo_likelihood_gradient
        else:
            self.model_selector = self.marginal_likelihood
            self.model_selector_gradient = self.marginal_likelihood_gradient

        if hyperpars is None:
            if optimizer not in ["bfgs", "diffev"]:
                optimizer = "bfgs"
                warn(msg.BAD_OPTIMIZER)
            if optimizer == "diffev":
                hyperpars = self.differential_evo()
            else:
                hyperpars = self.multistart_bfgs(n_processes=n_processes, starts=n_starts)

        self.set_hyperparameters(hyperpars)

    @staticmethod
    def _coerce_training_data(x, y):
        """Array conversion and shape checks of the constructor (regression.py:94-130): y must be
        one-dimensional, x (N, d) or — for d = 1 — anything that flattens to N values."""
        xa = np.asarray(x)
        ya = np.asarray(y).squeeze()
        if ya.ndim != 1:
            raise ValueError(msg.y_not_1d(ya.shape))
        if xa.ndim > 2:
            raise ValueError(msg.x_not_2d(xa.ndim, xa.shape))
        if xa.ndim < 2:
            xa = xa.reshape([xa.size, 1])
        if xa.shape[0] != ya.size:
            raise ValueError(msg.xy_mismatch(xa.shape, ya.size))
        return xa, ya

    # ---------------------------------------------------------------------------------
    # device plumbing
    # ---------------------------------------------------------------------------------
    @property
    def engine(self) -> GpEngine:
        if self._engine is None:
            self._engine = GpEngine(
                self.x, self.y, noise_var=self._noise_var, y_cov=self._y_cov, device=self._device,
                reserve=getattr(self, "_reserve", 0)
            )
        return self._engine

    def __getstate__(self):
        # device handles do not pickle (the reference pickles the regressor into worker
        # processes, regression.py:600-601); drop them and re-attach lazily
        state = self.__dict__.copy()
        state["_engine"] = None
        state["_K_cache"] = None
        state["_L_cache"] = None
        return state

    def _split_cov_theta(self, theta_cov):
        """(stationary-kernel parameters, WhiteNoise variance) of a covariance parameter vector.  With a
        HeteroscedasticNoise component the per-point variances exp(2 theta_i) are added to the data variances
        on the device (covariance.py:674-680) before the evaluation that follows."""
        theta_cov = np.asarray(theta_cov, dtype=float)
        extra = 0.0
        if self._wn_index is not None:
            extra = float(np.exp(2 * theta_cov[self._wn_index]))  # covariance.py:168
        if self._het_slice is not None:
            self.engine.set_noise(self._noise_total(theta_cov))
        return np.ascontiguousarray(theta_cov[self._stat_slice]), extra

    def _noise_total(self, theta_cov):
        base = np.zeros(self.n_points) if self._noise_var is None else self._noise_var
        return base + np.exp(2 * np.asarray(theta_cov, dtype=float)[self._het_slice])

    def _mean_at(self, pts):
        """Prior mean at the rows of `pts`: vectorised for the built-in means, point by point (as the reference
        does, regression.py:211) for user-defined ones."""
        if hasattr(self.mean, "at_points"):
            return self.mean.at_points(pts, self.mean_hyperpars)
        return array([self.mean(q, self.mean_hyperpars) for q in pts[:, None, :]])

    def _mix_args(self, theta_cp):
        """(kernel ids, sub-kernel parameter vectors, training-point weights) of a ChangePoint block."""
        kernels, thetas = self._mix.device_terms(theta_cp)
        return kernels, thetas, self._mix.weights(self._mix.x_cp, theta_cp)

    def _mix_window_terms(self, theta_cp):
        """Row-sum weights and window derivatives of a ChangePoint block with any number of regions.  The reference
        differentiates change-point c through the factor (1 - f_c) of K_c and the factor f_c of K_{c+1} only
        (covariance.py:588-593: `K_vals[i] * (A + A.T) + K_vals[i + 1] * (B + B.T)` with the BARE sub-kernel matrices),
        so 1/2 sum Q o dK/dphi_c = sum_i dphi f_c(i) (h_{c+1,0} - h_{c,1})(i) with the device row sums
        h_{m,0}(i) = sum_j Q_ij K_m,ij f_{m-1}(j),  h_{m,1}(i) = sum_j Q_ij K_m,ij (1 - f_m)(j).
        Returns (hw (nk, 2, n), [(dws of change-point c) ...])."""
        cp = self._mix
        nk = cp.n_kernels
        hw = zeros((nk, 2, cp.x_cp.size))
        dws = []
        for c_, slc in enumerate(cp.cp_slc):
            w, dw = cp.logistic_and_gradient(cp.x_cp, theta_cp[slc])
            hw[c_, 1] = 1.0 - w
            hw[c_ + 1, 0] = w
            dws.append(dw)
        return hw, dws

    @staticmethod
    def _mix_window_gradient(cp, dws, hrows, grad, scale=1.0):
        """grad[cp.cp_slc[c]] = scale * sum_i dphi f_c(i) (h_{c+1,0} - h_{c,1})(i) for every change-point c."""
        for c_, slc in enumerate(cp.cp_slc):
            dh = hrows[c_ + 1, 0] - hrows[c_, 1]
            grad[slc] = [scale * float((dw * dh).sum()) for dw in dws[c_]]

    def _refit_mixture_if_stale(self):
        # the fitted mixture's weights share a device buffer with the likelihood evaluations
        if self._mix is not None and getattr(self, "_mix_fit_stale", False):
            self.set_hyperparameters(self.hyperpars)

    @property
    def sig(self) -> ndarray:
        """Dense data-error covariance as in the reference (regression.py:320-322)."""
        if self._y_cov is not None:
            return self._y_cov
        if self._noise_var is not None:
            return np.diag(self._noise_var)
        return zeros([self.n_points, self.n_points])

    @property
    def K_xx(self) -> ndarray:
        if self._K_cache is None and self._generic:
            self._K_cache = self._dense_K(self.cov_hyperpars)
        if self._K_cache is None and self._mix is not None:
            # host composition of the sub-kernels' device builds (ChangePoint.build_covariance) + sig
            self._K_cache = self.cov.build_covariance(np.asarray(self.cov_hyperpars, dtype=float)) + self.sig
        if self._K_cache is None:
            if self._fit_noise is not None:  # a later evaluation may have left other per-point variances behind
                self.engine.set_noise(self._fit_noise)
            self._K_cache = self.engine.get_K()
        return self._K_cache

    @property
    def L(self) -> ndarray:
        if self._L_cache is None:
            self._refit_mixture_if_stale()  # a ChangePoint likelihood evaluation shares lane 0 with the fit
            self._L_cache = self.engine.get_L()
        return self._L_cache

    # ---------------------------------------------------------------------------------
    # public API (reference: regression.py:188-567)
    # ---------------------------------------------------------------------------------
    def __call__(self, points: ndarray):
        """Mean and standard deviation of the regression estimate at `points`
        (regression.py:188-216), evaluated as one batched device call."""
        p = self.process_points(points)
        if self._generic:
            return self._generic_predict(p)
        self._refit_mixture_if_stale()
        if self._mix is not None:
            theta_cp = np.asarray(self.cov_hyperpars, dtype=float)[self._stat_slice]
            gq = self._mix.weights(p[:, self._mix.axis], theta_cp)
            mu, neg = self.engine.predict_mix(p, gq)
            # K_qq[0, 0] = sum_m g_m(q)^2 a_m^2: cross-covariances carry neither jitter nor noise (covariance.py:529-544)
            amp2 = np.array([np.exp(2 * theta_cp[s][0]) for s in self._mix.cov_slc])
            var = (gq**2 * amp2[:, None]).sum(axis=0) + neg
        else:
            mu, var = self.engine.predict(p)
        return mu + self._mean_at(p), sqrt(abs(var))

    def set_hyperparameters(self, hyperpars: ndarray):
        """Update the hyper-parameters and re-fit (regression.py:218-244)."""
        if len(hyperpars) != self.n_hyperpars:
            raise ValueError(msg.wrong_hyperpar_count(self.n_hyperpars, len(hyperpars)))
        self.hyperpars = hyperpars
        self.mean_hyperpars = self.hyperpars[self.mean_slice]
        self.cov_hyperpars = self.hyperpars[self.cov_slice]
        self.mu = self.mean.build_mean(self.mean_hyperpars)
        if self._generic:
            self._K_cache = self._dense_K(self.cov_hyperpars)
            self._L_cache = None
            alpha, logdet, info = self.engine.fit_dense(self._K_cache, self.mu)
            if info != 0:
                raise LinAlgError("Matrix is not positive definite")
            self.alpha, self._logdet = alpha, logdet
            return
        theta_stat, extra = self._split_cov_theta(self.cov_hyperpars)
        self._fit_noise = self._noise_total(self.cov_hyperpars) if self._het_slice is not None else None
        self._K_cache = None
        self._L_cache = None
        if self._mix is not None:
            alpha, logdet, info = self.engine.fit_mix(*self._mix_args(theta_stat), extra, self.mu)
        else:
            alpha, logdet, info = self.engine.fit(self._kernel_id, theta_stat, extra, self.mu)
        self._mix_fit_stale = False
        if info != 0:
            raise LinAlgError("Matrix is not positive definite")  # numpy.linalg.cholesky, regression.py:241
        self.alpha = alpha
        self._logdet = logdet

    def add_point(self, x_new, y_new, y_err_new=None):
        """(extension) Append one training point and re-fit at the CURRENT hyper-parameters.

        With a SquaredExponential / RationalQuadratic kernel (optionally + WhiteNoise), diagonal data errors and
        free device capacity (`reserve`) this is an O(N^2) update of the fitted state: one new row of the Cholesky
        factor (a triangular sweep) and a fresh alpha (two sweeps) - `gpmi_append_point`.  In every other case
        the model is rebuilt and re-fitted at the same hyper-parameters (O(N^3), what the reference always does:
        optimisation.py:177-186 builds a new GpRegressor per added evaluation)."""
        x_new = np.asarray(x_new, dtype=float).reshape(1, self.n_dimensions)
        y_new = float(np.asarray(y_new).squeeze())
        if self._y_cov is not None:
            raise NotImplementedError("add_point with a dense y_cov: rebuild the regressor instead")
        if (self._noise_var is None) != (y_err_new is None):
            raise ValueError("y_err_new must be given exactly when the model was built with y_err")
        var_new = 0.0 if y_err_new is None else float(np.asarray(y_err_new).squeeze()) ** 2
        if self._het_slice is not None:
            # one noise hyper-parameter per training point: the hyper-parameter vector itself would have to grow
            raise NotImplementedError("add_point with HeteroscedasticNoise: rebuild the regressor with the enlarged data")
        fast = (not self._generic and self._mix is None and self._y_cov is None
                and self._engine is not None and self.engine.n < self.engine.capacity())
        x_all = np.vstack([self.x, x_new])
        y_all = np.append(self.y, y_new)
        if fast:
            # the device first: a pivot <= 0 (a duplicate point at fixed hyper-parameters) must leave host and device
            # state as they were.  The means are centred on the data, so the prior mean comes from a copy that already
            # knows the new point.
            import copy

            mean_new = copy.deepcopy(self.mean)
            mean_new.pass_spatial_data(x_all)
            mu_new = mean_new.build_mean(self.mean_hyperpars)
            alpha, logdet, info = self.engine.append_point(x_new[0], y_new, var_new, mu_new)
            if info != 0:
                raise LinAlgError("Matrix is not positive definite")
        self.x, self.y = x_all, y_all
        if self._noise_var is not None:
            self._noise_var = np.append(self._noise_var, var_new)
        self.n_points = self.y.size
        self.cov.pass_spatial_data(self.x)
        self.mean.pass_spatial_data(self.x)  # Linear / Quadratic means are centred on the data: all prior means move
        self._K_cache = self._L_cache = None
        if not fast:
            if self._engine is not None:
                self._engine.close()
            self._engine = None
            self._reserve = max(getattr(self, "_reserve", 0), 128)
            self.set_hyperparameters(self.hyperpars)
            return
        self.mu = mu_new
        self.alpha, self._logdet = alpha, logdet

    def check_error_data(self, y_err, y_cov):
        """Validate the data-error arguments (regression.py:246-322).  Returns
        (variance vector | None, dense covariance matrix | None) — the dense N x N `sig` of the
        reference is only materialised on request (the `sig` property)."""

        def as_array(value, keyword):
            if isinstance(value, (list, tuple)):
                return array(value).squeeze()
            if type(value) is not ndarray:
                raise TypeError(msg.not_an_array(keyword, ndarray, type(value)))
            return value

        n = self.n_points
        if y_cov is not None:
            y_cov = as_array(y_cov, "y_cov")
            if y_cov.shape != (n, n):
                raise ValueError(msg.Y_COV_SHAPE)
            if not (y_cov == y_cov.T).all():
                raise ValueError(msg.Y_COV_ASYMMETRIC)
            if y_err is not None:
                warn(msg.Y_ERR_AND_Y_COV)
            return None, np.ascontiguousarray(y_cov, dtype=float)
        if y_err is not None:
            y_err = as_array(y_err, "y_err")
            if y_err.shape != (n,):
                raise ValueError(msg.Y_ERR_SHAPE)
            return np.asarray(y_err, dtype=float) ** 2, None
        return None, None

    def process_points(self, points: ndarray) -> ndarray:
        """Bring query points to shape (M, d) (regression.py:324-349)."""
        q = np.asarray(points)
        d = self.n_dimensions
        if q.ndim > 2:
            raise ValueError(msg.points_not_2d(q.ndim, q.shape))
        if q.ndim <= 1:
            if d == 1:
                q = q.reshape([q.size, 1])  # a vector (or scalar) of 1-D positions
            elif q.ndim == 1 and q.size == d:
                q = q.reshape([1, d])       # one d-dimensional position
        if q.shape[1] != d:
            raise ValueError(msg.points_wrong_width(d, q.shape))
        return q

    def _require_gradient_terms(self):
        if self._generic:
            return
        if self._mix is not None:
            self._mix.gradient_terms(None, None, None)  # ChangePoint has none either (covariance.py:38-44)
        if self._kernel_id != 0:
            # RationalQuadratic has no gradient_terms (covariance.py:38-44): same error as the reference
            self._stat.gradient_terms(None, None, None)

    def gradient(self, points: ndarray):
        """Mean and covariance of the gradient of the estimate (regression.py:351-385)."""
        self._require_gradient_terms()
        p = self.process_points(points)
        if self._generic:
            return self._generic_gradient(p)
        gmu, gcov = self.engine.gradient(p)
        return gmu.squeeze(), gcov.squeeze()

    def spatial_derivatives(self, points: ndarray):
        """Gradients of the predictive mean and variance (regression.py:387-419)."""
        self._require_gradient_terms()
        p = self.process_points(points)
        if self._generic:
            return self._generic_spatial_derivatives(p)
        dmu, dvar = self.engine.spatial_derivatives(p)
        return dmu.squeeze(), dvar.squeeze()

    def spatial_derivatives_batch(self, points: ndarray):
        """(extension) `spatial_derivatives` for M points with un-squeezed (M, d) results: the form the batched
        acquisition functions consume."""
        self._require_gradient_terms()
        p = self.process_points(points)
        if self._generic:
            dmu, dvar = self._generic_spatial_derivatives(p)
            return dmu.reshape(len(p), -1), dvar.reshape(len(p), -1)
        return self.engine.spatial_derivatives(p)

    def build_posterior(self, points: ndarray, mean_only=False):
        """Posterior mean vector and covariance matrix (regression.py:421-449)."""
        v = self.process_points(points)
        if self._generic:
            return self._generic_posterior(v, mean_only)
        self._refit_mixture_if_stale()
        if self._mix is not None:
            theta_cp = np.asarray(self.cov_hyperpars, dtype=float)[self._stat_slice]
            mu, sigma = self.engine.posterior_mix(v, self._mix.weights(v[:, self._mix.axis], theta_cp), mean_only=mean_only)
        else:
            mu, sigma = self.engine.posterior(v, mean_only=mean_only)
        mu = mu + self._mean_at(v)
        if mean_only:
            return mu
        return mu, sigma

    def loo_predictions(self):
        """Leave-one-out predictions, R&W eq. 5.12 (regression.py:451-466)."""
        if not self._generic:
            self._refit_mixture_if_stale()
        var = 1.0 / self.engine.loo_diag()
        return self.y - self.alpha * var, sqrt(var)

    def loo_likelihood(self, theta: ndarray) -> float:
        """Leave-one-out log-likelihood, R&W eqs. 5.10-5.12 (regression.py:468-487)."""
        theta = np.asarray(theta, dtype=float)
        if self._generic:
            return self._generic_loo(theta, want_gradient=False)
        theta_stat, extra = self._split_cov_theta(theta[self.cov_slice])
        mu = self.mean.build_mean(theta[self.mean_slice])
        if self._mix is not None:
            alpha, ikdiag, info = self.engine.loo_terms_mix(*self._mix_args(theta_stat), extra, mu)
            self._mix_fit_stale = True
        else:
            alpha, ikdiag, info = self.engine.loo_terms(self._kernel_id, theta_stat, extra, mu)
        if info != 0:
            warn("Cholesky decomposition failure in loo_likelihood")
            return -1e50
        var = 1.0 / ikdiag
        return float(-0.5 * (var * alpha**2 + np.log(var)).sum())

    def loo_likelihood_gradient(self, theta: ndarray):
        """LOO log-likelihood and its gradient, R&W eqs. 5.10-5.14 (regression.py:489-526)."""
        theta = np.asarray(theta, dtype=float)
        if self._generic:
            return self._generic_loo(theta, want_gradient=True)
        if self._mix is not None or self._het_slice is not None:
            if self._loo_batch_ok():
                # (round 5) the lockstep kernels take a batch of one: gpmi_loo_grad_batch_mix / gpmi_loo_grad_batch_noise
                vals, grads = self.loo_likelihood_gradient_batch(theta[None, :])
                return float(vals[0]), grads[0]
            # beyond the lockstep sizes, or a mixture with per-point noise on top: the device factorises / inverts the
            # dense K (gpmi_loo_dense) and each component of the gradient is an O(N^2) contraction with that component's
            # own derivative
            return self._dense_loo_gradient(theta)
        theta_stat, extra = self._split_cov_theta(theta[self.cov_slice])
        mu, grad_mu = self.mean.mean_and_gradients(theta[self.mean_slice])
        alpha, ikdiag, pvec, g_stat, trace_q, info = self.engine.loo_grad(self._kernel_id, theta_stat, extra, mu)
        if info != 0:
            raise LinAlgError("Matrix is not positive definite")  # regression.py:501 has no guard
        var = 1.0 / ikdiag
        LOO = float(-0.5 * (var * alpha**2 + np.log(var)).sum())
        grad = zeros(self.n_hyperpars)
        # sum(c1 * (iK @ dmu)) = (iK @ c1) . dmu   (regression.py:516-520)
        grad[self.mean_slice] = array([(pvec * dmu).sum() for dmu in grad_mu])
        g_cov = zeros(self.cov.n_params)
        g_cov[self._stat_slice] = g_stat
        if self._wn_index is not None:
            g_cov[self._wn_index] = 2.0 * extra * trace_q  # dK = 2 sigma^2 I (covariance.py:171-175)
        grad[self.cov_slice] = g_cov
        return LOO, grad

    def _loo_batch_ok(self):
        """A ChangePoint mixture or a HeteroscedasticNoise model (not both) whose leave-one-out gradient the lockstep
        kernels serve: lockstep sizes, diagonal data errors."""
        return ((self._mix is None) != (self._het_slice is None) and not self._generic and self._y_cov is None
                and self.engine.capacity() <= 4096)

    def loo_likelihood_gradient_batch(self, thetas: ndarray):
        """(extension) `loo_likelihood_gradient` for T hyper-parameter vectors in one device call (gpmi_loo_grad_batch:
        for N <= 4096 the evaluations advance in lockstep): returns (LOO (T,), grad (T, P)).  What the lockstep
        multi-start search evaluates per round when the model selector is the cross-validation objective
        (regression.py:159-164)."""
        thetas = np.atleast_2d(np.asarray(thetas, dtype=float))
        het_batch = self._het_slice is not None and self._loo_batch_ok()  # (round 5: gpmi_loo_grad_batch_noise)
        if self._mix is not None and self._loo_batch_ok():
            return self._mixture_loo_gradient_batch(thetas)  # (round 5: gpmi_loo_grad_batch_mix)
        if self._generic or self._mix is not None or (self._het_slice is not None and not het_batch):
            res = [self.loo_likelihood_gradient(t) for t in thetas]
            return np.array([r[0] for r in res]), np.array([r[1] for r in res])
        split = [self._split_cov_theta(t[self.cov_slice]) for t in thetas]
        th = np.array([s_[0] for s_ in split])
        ex = np.array([s_[1] for s_ in split])
        means = [self.mean.mean_and_gradients(t[self.mean_slice]) for t in thetas]
        mean_kw = (dict(mu_const=thetas[:, 0]) if isinstance(self.mean, ConstantMean)
                   else dict(mus=np.array([m[0] for m in means])))
        mdiag = None
        if het_batch:
            # HeteroscedasticNoise: every evaluation has noise variances of its own
            noise = np.array([self._noise_total(t[self.cov_slice]) for t in thetas])
            alpha, ikdiag, pvec, g_stat, trace_q, mdiag, info = self.engine.loo_grad_batch(self._kernel_id, th, ex,
                                                                                           noise_var=noise, **mean_kw)
        else:
            alpha, ikdiag, pvec, g_stat, trace_q, info = self.engine.loo_grad_batch(self._kernel_id, th, ex, **mean_kw)
        if (info != 0).any():
            raise LinAlgError("Matrix is not positive definite")  # regression.py:501 has no guard
        values = np.empty(len(thetas))
        grads = zeros((len(thetas), self.n_hyperpars))
        for t in range(len(thetas)):
            var = 1.0 / ikdiag[t]
            values[t] = float(-0.5 * (var * alpha[t] ** 2 + np.log(var)).sum())
            grads[t, self.mean_slice] = array([(pvec[t] * dmu).sum() for dmu in means[t][1]])
            g_cov = zeros(self.cov.n_params)
            g_cov[self._stat_slice] = g_stat[t]
            if self._wn_index is not None:
                g_cov[self._wn_index] = 2.0 * ex[t] * trace_q[t]  # dK = 2 sigma^2 I (covariance.py:171-175)
            if mdiag is not None:
                # dK / d ln sigma_i = 2 sigma_i^2 e_i e_i^T (covariance.py:683-689): c1.(Z alpha) - c2.diag(Z K^-1) with
                # Z = K^-1 dK collapses to 2 sigma_i^2 (p_i alpha_i - M_ii), M = K^-1 diag(c2) K^-1
                g_cov[self._het_slice] = (2.0 * np.exp(2 * thetas[t][self.cov_slice][self._het_slice])
                                          * (pvec[t] * alpha[t] - mdiag[t]))
            grads[t, self.cov_slice] = g_cov
        return values, grads

    def marginal_likelihood(self, theta: ndarray) -> float:
        """Log-marginal likelihood, R&W eq. 5.8 without the 2 pi constant (regression.py:528-542)."""
        theta = np.asarray(theta, dtype=float)
        if self._generic:
            value, _, _, info = self.engine.lml_dense(self._dense_K(theta[self.cov_slice]),
                                                      self.mean.build_mean(theta[self.mean_slice]))
            if info != 0:
                warn("Cholesky decomposition failure in marginal_likelihood")
                return -1e50
            return float(value)
        theta_stat, extra = self._split_cov_theta(theta[self.cov_slice])
        mu = self.mean.build_mean(theta[self.mean_slice])
        if self._mix is not None:
            value, info = self.engine.lml_mix(*self._mix_args(theta_stat), extra, mu)
            self._mix_fit_stale = True
        else:
            value, info = self.engine.lml(self._kernel_id, theta_stat, extra, mu)
        if info != 0:
            warn("Cholesky decomposition failure in marginal_likelihood")
            return -1e50
        return float(value)

    def batch_independent_values(self, on: bool = True):
        """(extension) Make `marginal_likelihood_batch` values independent of the batch they are evaluated in
        (`GPMI_OPT_LOCKSTEP_ALWAYS`): every call, also of a single theta, takes the lockstep device path.  The
        lockstep MCMC drivers switch this on so that a chain's trajectory does not depend on which other chains
        happen to share its proposal rounds."""
        from inference_amd import _lib

        self.engine.set_option(_lib.OPT_LOCKSTEP_ALWAYS, 1 if on else 0)
        self._batch_independent = bool(on)

    def marginal_likelihood_batch(self, thetas: ndarray) -> ndarray:
        """(extension) `marginal_likelihood` for T hyper-parameter vectors at once, spread over
        the device's worker streams — the unit the grid sweep / PT driver shards over GPUs."""
        thetas = np.asarray(thetas, dtype=float)
        if thetas.size == 0:
            return np.empty(0)  # an empty shard (more ranks than evaluations)
        thetas = np.atleast_2d(thetas)
        if self._generic or self._het_slice is not None or self._mix is not None:  # per-point terms change with theta: one at a time
            return np.array([self.marginal_likelihood(t) for t in thetas])
        split = [self._split_cov_theta(t[self.cov_slice]) for t in thetas]
        th = np.array([s[0] for s in split])
        ex = np.array([s[1] for s in split])
        if isinstance(self.mean, ConstantMean):
            vals, info = self.engine.lml_batch(self._kernel_id, th, ex, mu_const=thetas[:, 0])
        else:
            mus = np.array([self.mean.build_mean(t[self.mean_slice]) for t in thetas])
            vals, info = self.engine.lml_batch(self._kernel_id, th, ex, mus=mus)
        if (info != 0).any():
            warn("Cholesky decomposition failure in marginal_likelihood")
        return vals

    def async_batches(self):
        """(extension) Can `marginal_likelihood_batch_submit` / `_wait` serve this model?  (A device kernel, a model small
        enough for lockstep batches, diagonal data errors: the cases `gpmi_lml_batch_submit` takes.)"""
        return (not self._generic and self._het_slice is None and self._mix is None and self._y_cov is None
                and self.engine.capacity() <= 4096)

    def marginal_likelihood_batch_submit(self, thetas: ndarray, slot: int):
        """(extension) Start `marginal_likelihood_batch(thetas)` in slot 0 or 1 and return at once; at most
        `engine.ASYNC_MAX` vectors per slot.  The two slots run side by side on the device: a driver with two groups of
        chains does the bookkeeping of one group while the other group's likelihoods are being evaluated."""
        thetas = np.atleast_2d(np.asarray(thetas, dtype=float))
        split = [self._split_cov_theta(t[self.cov_slice]) for t in thetas]
        th = np.array([s[0] for s in split])
        ex = np.array([s[1] for s in split])
        if isinstance(self.mean, ConstantMean):
            self.engine.lml_batch_submit(slot, self._kernel_id, th, ex, mu_const=thetas[:, 0])
        else:
            mus = np.array([self.mean.build_mean(t[self.mean_slice]) for t in thetas])
            self.engine.lml_batch_submit(slot, self._kernel_id, th, ex, mus=mus)

    def marginal_likelihood_batch_wait(self, slot: int) -> ndarray:
        """(extension) The values of the batch submitted in `slot` (same values as `marginal_likelihood_batch`)."""
        vals, info = self.engine.lml_batch_wait(slot)
        if (info != 0).any():
            warn("Cholesky decomposition failure in marginal_likelihood")
        return vals

    def marginal_likelihood_gradient(self, theta: ndarray):
        """LML and its gradient, R&W eqs. 5.8-5.9 (regression.py:544-567)."""
        theta = np.asarray(theta, dtype=float)
        if self._generic:
            return self._generic_lml_gradient(theta)
        theta_stat, extra = self._split_cov_theta(theta[self.cov_slice])
        mu, grad_mu = self.mean.mean_and_gradients(theta[self.mean_slice])
        if self._mix is not None:
            lml, g_stat, alpha, trace_q, info = self._mixture_gradient(theta_stat, extra, mu)
        else:
            lml, g_stat, trace_q, alpha, info = self.engine.lml_grad(self._kernel_id, theta_stat, extra, mu)
        if info != 0:
            raise LinAlgError("Matrix is not positive definite")  # regression.py:555 has no guard
        grad = zeros(self.n_hyperpars)
        grad[self.mean_slice] = array([(alpha * dmu).sum() for dmu in grad_mu])
        g_cov = zeros(self.cov.n_params)
        g_cov[self._stat_slice] = g_stat
        if self._wn_index is not None:
            g_cov[self._wn_index] = extra * trace_q  # 1/2 sum Q o (2 sigma^2 I), covariance.py:171-175
        if self._het_slice is not None:
            # dK/d ln sigma_i = 2 sigma_i^2 e_i e_i^T (covariance.py:682-686): 1/2 Q_ii 2 sigma_i^2
            g_cov[self._het_slice] = np.exp(2 * theta[self.cov_slice][self._het_slice]) * self.engine.lml_grad_qdiag()
        grad[self.cov_slice] = g_cov
        return lml, grad

    def marginal_likelihood_gradient_batch(self, thetas: ndarray):
        """(extension) `marginal_likelihood_gradient` for T hyper-parameter vectors in one device call
        (gpmi_lml_grad_batch: for N <= 4096 the evaluations advance in lockstep, every launch carrying all of them):
        returns (lml (T,), grad (T, P)).  What the lockstep multi-start search evaluates per round."""
        thetas = np.atleast_2d(np.asarray(thetas, dtype=float))
        if self._mix is not None and not self._generic and self._het_slice is None:
            return self._mixture_gradient_batch(thetas)  # (round 4: gpmi_lml_grad_batch_mix; any number of regions: round 5)
        # a mixture with per-point noise on top, or per-point noise beside a dense y_cov (gpmi_lml_grad_batch_noise
        # carries diagonals only - the condition of _lockstep_search): one at a time
        if self._generic or self._mix is not None or (self._het_slice is not None and self._y_cov is not None):
            res = [self.marginal_likelihood_gradient(t) for t in thetas]
            return np.array([r[0] for r in res]), np.array([r[1] for r in res])
        th = np.array([np.ascontiguousarray(t[self.cov_slice][self._stat_slice]) for t in thetas])
        ex = np.array([float(np.exp(2 * t[self.cov_slice][self._wn_index])) if self._wn_index is not None else 0.0
                       for t in thetas])
        means = [self.mean.mean_and_gradients(t[self.mean_slice]) for t in thetas]
        mean_kw = (dict(mu_const=thetas[:, 0]) if isinstance(self.mean, ConstantMean)
                   else dict(mus=np.array([m[0] for m in means])))
        qdiag = None
        if self._het_slice is not None:
            # HeteroscedasticNoise: every evaluation has noise variances of its own (round 4: gpmi_lml_grad_batch_noise)
            noise = np.array([self._noise_total(t[self.cov_slice]) for t in thetas])
            lml, g_stat, trace_q, alpha, qdiag, info = self.engine.lml_grad_batch_noise(self._kernel_id, th, ex, noise,
                                                                                        **mean_kw)
        else:
            lml, g_stat, trace_q, alpha, info = self.engine.lml_grad_batch(self._kernel_id, th, ex, **mean_kw)
        if (info != 0).any():
            raise LinAlgError("Matrix is not positive definite")  # regression.py:555 has no guard
        grads = zeros((len(thetas), self.n_hyperpars))
        for t in range(len(thetas)):
            grads[t, self.mean_slice] = array([(alpha[t] * dmu).sum() for dmu in means[t][1]])
            g_cov = zeros(self.cov.n_params)
            g_cov[self._stat_slice] = g_stat[t]
            if self._wn_index is not None:
                g_cov[self._wn_index] = ex[t] * trace_q[t]
            if qdiag is not None:  # dK/d ln sigma_i = 2 sigma_i^2 e_i e_i^T (covariance.py:682-686)
                g_cov[self._het_slice] = np.exp(2 * thetas[t][self.cov_slice][self._het_slice]) * qdiag[t]
            grads[t, self.cov_slice] = g_cov
        return lml, grads

    def _mixture_gradient(self, theta_cp, extra, mu):
        """LML gradient with respect to a ChangePoint block (covariance.py:561-594): the sub-kernels' parameters
        from the device contraction on the weight-scaled inverse, the window parameters from the device row
        sums h_m(i) = sum_j Q_ij K_m,ij g_m(j) contracted here with d g_m / d phi."""
        cp = self._mix
        kernels, thetas, g = self._mix_args(theta_cp)
        hw, dws = self._mix_window_terms(theta_cp)
        lml, g_sub, hrows, alpha, info = self.engine.lml_grad_mix(kernels, thetas, g, extra, mu, row_weights=hw)
        self._mix_fit_stale = True
        grad = zeros(cp.n_params)
        grad[: g_sub.size] = g_sub
        # 1/2 sum Q o (K_c o (A + A^T) + K_{c+1} o (B + B^T)), A = -dw (1 - w)^T, B = dw w^T  =  sum_i dw_i (h_{c+1,0} - h_{c,1})_i
        self._mix_window_gradient(cp, dws, hrows, grad)
        trace_q = float(self.engine.lml_grad_qdiag().sum()) if self._wn_index is not None else 0.0
        return lml, grad, alpha, trace_q, info

    def _mixture_gradient_batch(self, thetas):
        """`marginal_likelihood_gradient` of a ChangePoint model (any number of regions) for T hyper-parameter vectors in one device
        call: per evaluation the window weights (O(N) on the host) go along with the sub-kernels' parameters, the device
        returns the sub-kernel gradients and the row sums h_m, and the window parameters' gradient is contracted here
        as in `_mixture_gradient` (covariance.py:561-594)."""
        cp = self._mix
        T = len(thetas)
        stat = [np.ascontiguousarray(t[self.cov_slice][self._stat_slice]) for t in thetas]
        ex = np.array([float(np.exp(2 * t[self.cov_slice][self._wn_index])) if self._wn_index is not None else 0.0
                       for t in thetas])
        means = [self.mean.mean_and_gradients(t[self.mean_slice]) for t in thetas]
        mean_kw = (dict(mu_const=thetas[:, 0]) if isinstance(self.mean, ConstantMean)
                   else dict(mus=np.array([m[0] for m in means])))
        args = [self._mix_args(s_) for s_ in stat]
        kernels = args[0][0]
        win = [self._mix_window_terms(s_) for s_ in stat]
        lml, g_sub, hrows, alpha, qdiag, info = self.engine.lml_grad_batch_mix(
            kernels, [a[1] for a in args], np.array([a[2] for a in args]), ex, want_qdiag=self._wn_index is not None,
            row_weights=np.array([w_[0] for w_ in win]), **mean_kw)
        self._mix_fit_stale = True
        if (info != 0).any():
            raise LinAlgError("Matrix is not positive definite")  # regression.py:555 has no guard
        grads = zeros((T, self.n_hyperpars))
        for t in range(T):
            g_cp = zeros(cp.n_params)
            g_cp[: g_sub.shape[1]] = g_sub[t]
            self._mix_window_gradient(cp, win[t][1], hrows[t], g_cp)
            grads[t, self.mean_slice] = array([(alpha[t] * dmu).sum() for dmu in means[t][1]])
            g_cov = zeros(self.cov.n_params)
            g_cov[self._stat_slice] = g_cp
            if self._wn_index is not None:
                g_cov[self._wn_index] = ex[t] * float(qdiag[t].sum())
            grads[t, self.cov_slice] = g_cov
        return lml, grads

    def _mixture_loo_gradient_batch(self, thetas):
        """`loo_likelihood_gradient` of a ChangePoint model (any number of regions) for T hyper-parameter vectors in one device call
        (gpmi_loo_grad_batch_mix).  With Q = sym(p alpha^T) - K^-1 diag(c2) K^-1 the gradient of regression.py:509-514 is
        sum Q o dK_j for every component: the sub-kernels' from the device contraction on the weight-scaled matrix, the
        window parameters' from the device row sums h_m contracted here with d g_m / d phi as in `_mixture_gradient`
        (doubled: the likelihood gradient carries a 1/2 that this one does not), WhiteNoise's 2 s^2 trace(Q)."""
        cp = self._mix
        T = len(thetas)
        stat = [np.ascontiguousarray(t[self.cov_slice][self._stat_slice]) for t in thetas]
        ex = np.array([float(np.exp(2 * t[self.cov_slice][self._wn_index])) if self._wn_index is not None else 0.0
                       for t in thetas])
        means = [self.mean.mean_and_gradients(t[self.mean_slice]) for t in thetas]
        mean_kw = (dict(mu_const=thetas[:, 0]) if isinstance(self.mean, ConstantMean)
                   else dict(mus=np.array([m[0] for m in means])))
        args = [self._mix_args(s_) for s_ in stat]
        win = [self._mix_window_terms(s_) for s_ in stat]
        alpha, ikdiag, pvec, mdiag, g_sub, hrows, info = self.engine.loo_grad_batch_mix(
            args[0][0], [a[1] for a in args], np.array([a[2] for a in args]), ex,
            row_weights=np.array([w_[0] for w_ in win]), **mean_kw)
        self._mix_fit_stale = True
        if (info != 0).any():
            raise LinAlgError("Matrix is not positive definite")  # regression.py:501 has no guard
        values = np.empty(T)
        grads = zeros((T, self.n_hyperpars))
        for t in range(T):
            var = 1.0 / ikdiag[t]
            values[t] = float(-0.5 * (var * alpha[t] ** 2 + np.log(var)).sum())
            g_cp = zeros(cp.n_params)
            g_cp[: g_sub.shape[1]] = g_sub[t]
            self._mix_window_gradient(cp, win[t][1], hrows[t], g_cp, scale=2.0)
            grads[t, self.mean_slice] = array([(pvec[t] * dmu).sum() for dmu in means[t][1]])
            g_cov = zeros(self.cov.n_params)
            g_cov[self._stat_slice] = g_cp
            if self._wn_index is not None:
                g_cov[self._wn_index] = 2.0 * ex[t] * float((pvec[t] * alpha[t] - mdiag[t]).sum())
            grads[t, self.cov_slice] = g_cov
        return values, grads

    # ---------------------------------------------------------------------------------
    # covariance functions that only implement the plugin ABC: the plugin's host methods make the dense
    # matrices, the device factorises / solves (gpmi_fit_dense, gpmi_lml_dense, gpmi_predict_dense, gpmi_solve_rows)
    # ---------------------------------------------------------------------------------
    def _dense_K(self, theta_cov):
        return np.ascontiguousarray(self.cov.build_covariance(np.asarray(theta_cov, dtype=float)) + self.sig)

    def _generic_cross(self, p):
        """(K_qx (M, N), diag K_qq (M,)) from the plugin's __call__ (regression.py:209-210 evaluates it point by point)."""
        th = np.asarray(self.cov_hyperpars, dtype=float)
        Kq = np.ascontiguousarray(self.cov(p, self.x, th))
        kqq = array([float(self.cov(q, q, th)[0, 0]) for q in p[:, None, :]])
        return Kq, kqq

    def _generic_predict(self, p):
        Kq, kqq = self._generic_cross(p)
        ka, ss = self.engine.predict_dense(Kq)
        return ka + self._mean_at(p), sqrt(abs(kqq - ss))

    def _generic_posterior(self, v, mean_only):
        th = np.asarray(self.cov_hyperpars, dtype=float)
        Kq = np.ascontiguousarray(self.cov(v, self.x, th))
        mu = self.engine.predict_dense(Kq, want_var=False)[0] + self._mean_at(v)
        if mean_only:
            return mu
        _, gram = self.engine.solve_rows(Kq, want_rows=False, want_gram=True)
        return mu, self.cov(v, v, th) - gram  # regression.py:447-448

    def _generic_lml_gradient(self, theta):
        """regression.py:544-567 with K^-1 from the device and dK from the plugin's covariance_and_gradients."""
        K, grad_K = self.cov.covariance_and_gradients(theta[self.cov_slice])
        mu, grad_mu = self.mean.mean_and_gradients(theta[self.mean_slice])
        lml, alpha, iK, info = self.engine.lml_dense(np.ascontiguousarray(K + self.sig), mu, want_alpha=True,
                                                     want_inverse=True)
        if info != 0:
            raise LinAlgError("Matrix is not positive definite")
        Q = alpha[:, None] * alpha[None, :] - iK
        grad = zeros(self.n_hyperpars)
        grad[self.mean_slice] = array([(alpha * dmu).sum() for dmu in grad_mu])
        grad[self.cov_slice] = array([0.5 * (Q * dK.T).sum() for dK in grad_K])
        return lml, grad

    def _generic_loo(self, theta, want_gradient):
        """regression.py:468-526.  The device returns alpha, diag(K^-1) and - for the gradient - the two
        parameter-independent pieces p = K^-1 c1 and W = K^-1 diag(c2) K^-1, so that each of the P gradient
        components is an O(N^2) contraction with the plugin's own dK_j instead of the reference's N^3 product
        K^-1 dK_j (:512): sum_i c1_i (K^-1 dK alpha)_i - c2_i (K^-1 dK K^-1)_ii = p . (dK alpha) - sum dK o W."""
        if want_gradient:
            K, grad_K = self.cov.covariance_and_gradients(theta[self.cov_slice])
            mu, grad_mu = self.mean.mean_and_gradients(theta[self.mean_slice])
        else:
            K, mu = self.cov.build_covariance(theta[self.cov_slice]), self.mean.build_mean(theta[self.mean_slice])
        alpha, ikdiag, pvec, W, info = self.engine.loo_dense(np.ascontiguousarray(K + self.sig), mu, want_gradient)
        if info != 0:
            if want_gradient:
                raise LinAlgError("Matrix is not positive definite")
            warn("Cholesky decomposition failure in loo_likelihood")
            return -1e50
        var = 1.0 / ikdiag
        LOO = float(-0.5 * (var * alpha**2 + np.log(var)).sum())
        if not want_gradient:
            return LOO
        grad = zeros(self.n_hyperpars)
        grad[self.mean_slice] = array([(pvec * dmu).sum() for dmu in grad_mu])
        grad[self.cov_slice] = array([pvec @ (dK @ alpha) - (dK * W).sum() for dK in grad_K])
        return LOO, grad

    def _component_gradients(self, theta_cov, contract, diag_terms):
        """Gradient of an objective with respect to every covariance parameter from per-component pieces:
        `contract(dK)` for a component with dense derivative matrices (its own covariance_and_gradients),
        `diag_terms` = the per-point quantity t_i with d objective / d K_ii = t_i for the diagonal components
        (WhiteNoise: 2 sigma^2 sum t_i, covariance.py:171-175; HeteroscedasticNoise: 2 sigma_i^2 t_i, :682-686)."""
        from inference_amd.gp.covariance import CompositeCovariance, HeteroscedasticNoise, WhiteNoise

        comps = list(zip(self.cov.components, self.cov.slices)) if isinstance(self.cov, CompositeCovariance) \
            else [(self.cov, slice(0, self.cov.n_params))]
        grad = zeros(self.cov.n_params)
        for comp, slc in comps:
            th = theta_cov[slc]
            if isinstance(comp, HeteroscedasticNoise):
                grad[slc] = 2.0 * np.exp(2 * th) * diag_terms
            elif isinstance(comp, WhiteNoise):
                grad[slc] = 2.0 * np.exp(2 * th[0]) * diag_terms.sum()
            else:
                grad[slc] = [contract(dK) for dK in comp.covariance_and_gradients(th)[1]]
        return grad

    def _dense_lml_gradient(self, theta):
        """regression.py:544-567 for covariance objects without a fused gradient kernel: K(theta) from the object's own
        build (device kernels composed on the host), K^-1 and alpha from the device (gpmi_lml_dense)."""
        theta_cov = theta[self.cov_slice]
        mu, grad_mu = self.mean.mean_and_gradients(theta[self.mean_slice])
        lml, alpha, iK, info = self.engine.lml_dense(self._dense_K(theta_cov), mu, want_alpha=True, want_inverse=True)
        if info != 0:
            raise LinAlgError("Matrix is not positive definite")
        Q = alpha[:, None] * alpha[None, :] - iK
        grad = zeros(self.n_hyperpars)
        grad[self.mean_slice] = array([(alpha * dmu).sum() for dmu in grad_mu])
        grad[self.cov_slice] = self._component_gradients(theta_cov, lambda dK: 0.5 * (Q * dK.T).sum(), 0.5 * np.diag(Q))
        return lml, grad

    def _dense_loo_gradient(self, theta):
        """regression.py:489-526 for the same objects: alpha, diag(K^-1), p = K^-1 c1 and W = K^-1 diag(c2) K^-1
        from the device (gpmi_loo_dense), d LOO / d theta_j = p . (dK_j alpha) - sum dK_j o W on the host."""
        theta_cov = theta[self.cov_slice]
        mu, grad_mu = self.mean.mean_and_gradients(theta[self.mean_slice])
        alpha, ikdiag, pvec, W, info = self.engine.loo_dense(self._dense_K(theta_cov), mu, True)
        if info != 0:
            raise LinAlgError("Matrix is not positive definite")
        var = 1.0 / ikdiag
        LOO = float(-0.5 * (var * alpha**2 + np.log(var)).sum())
        grad = zeros(self.n_hyperpars)
        grad[self.mean_slice] = array([(pvec * dmu).sum() for dmu in grad_mu])
        grad[self.cov_slice] = self._component_gradients(
            theta_cov, lambda dK: pvec @ (dK @ alpha) - (dK * W).sum(), pvec * alpha - np.diag(W))
        return LOO, grad

    def _gradient_pieces(self, p):
        """Per query point: A (d, N), R, k (N,) from the plugin's gradient_terms / __call__ (regression.py:368-372)."""
        th = np.asarray(self.cov_hyperpars, dtype=float)
        Kq = np.ascontiguousarray(self.cov(p, self.x, th))
        terms = [self.cov.gradient_terms(q, self.x, th) for q in p]
        return Kq, [t[0] for t in terms], [t[1] for t in terms]

    def _generic_gradient(self, p):
        """regression.py:351-385: mean and covariance of the gradient, the triangular solves batched on the device."""
        Kq, As, Rs = self._gradient_pieces(p)
        d = self.n_dimensions
        gmu = array([A @ (k * self.alpha) for A, k in zip(As, Kq)])
        rhs = np.concatenate([A * k[None, :] for A, k in zip(As, Kq)], axis=0)  # (M d, N)
        gcov = np.empty((len(p), d, d))
        step = max(1, 4096 // d)
        for lo in range(0, len(p), step):
            hi = min(lo + step, len(p))
            X, _ = self.engine.solve_rows(rhs[lo * d:hi * d])
            for i in range(lo, hi):
                Xi = X[(i - lo) * d:(i - lo + 1) * d]
                gcov[i] = Rs[i] - Xi @ Xi.T  # regression.py:379 (a vector R broadcasts over the rows, as there)
        return gmu.squeeze(), gcov.squeeze()

    def _generic_spatial_derivatives(self, p):
        """regression.py:387-419: grad mu = A (k o alpha), grad var = -2 (A o k) K^-1 k."""
        Kq, As, _ = self._gradient_pieces(p)
        d = self.n_dimensions
        dmu = array([A @ (k * self.alpha) for A, k in zip(As, Kq)])
        dvar = np.empty((len(p), d))
        step = max(1, 4096 // (d + 1))
        for lo in range(0, len(p), step):
            hi = min(lo + step, len(p))
            rows = np.concatenate([np.vstack([As[i] * Kq[i][None, :], Kq[i][None, :]]) for i in range(lo, hi)], axis=0)
            X, _ = self.engine.solve_rows(rows)  # rows L^-T: (A o k) L^-T and k L^-T per point
            for i in range(lo, hi):
                blk = X[(i - lo) * (d + 1):(i - lo + 1) * (d + 1)]
                dvar[i] = -2.0 * blk[:d] @ blk[d]
        return dmu.squeeze(), dvar.squeeze()

    # ---------------------------------------------------------------------------------
    # hyper-parameter search (regression.py:569-605): SciPy drivers on the host, every
    # objective evaluation on the device
    # ---------------------------------------------------------------------------------
    def differential_evo(self) -> ndarray:
        """regression.py:569-573.  The default is the reference's call: SciPy's `updating="immediate"` population walk, one
        objective evaluation per call (each a latency-bound launch train at small N), the reference's trajectory under the
        same seed.  `diffev_batched=True` (extension, opt-in: another - equally valid - trajectory) hands SciPy a vectorised
        objective with `updating="deferred"`: a whole generation (15 x P trial vectors) is ONE lockstep device call
        (`marginal_likelihood_batch`; the leave-one-out objective through `loo_likelihood_gradient_batch`)."""
        if getattr(self, "_diffev_batched", False):
            def neg_generation(pop):  # SciPy: (P, S) -> (S,)
                return -self.model_selector_batch(np.ascontiguousarray(pop.T))

            opt_result = differential_evolution(func=neg_generation, bounds=self.hp_bounds, vectorized=True, updating="deferred")
            self.search_log = [(None, array(opt_result.x), float(opt_result.fun))]
            return opt_result.x
        opt_result = differential_evolution(func=lambda t: -self.model_selector(t), bounds=self.hp_bounds)
        return opt_result.x

    def model_selector_batch(self, thetas: ndarray) -> ndarray:
        """(extension) `model_selector` for T hyper-parameter vectors: one device call where batched kernels exist (the
        marginal likelihood of a plain kernel: gpmi_lml_batch; the leave-one-out likelihood at lockstep sizes:
        gpmi_loo_grad_batch), else one evaluation after another.  A failed factorisation scores -1e50 as in the reference
        (regression.py:540-542, 484-487)."""
        thetas = np.atleast_2d(np.asarray(thetas, dtype=float))
        if self.model_selector == self.marginal_likelihood:
            with catch_warnings():
                simplefilter("ignore")  # (one "Cholesky decomposition failure" warning per generation says nothing new)
                return np.asarray(self.marginal_likelihood_batch(thetas), dtype=float)
        if self._lockstep_search():
            try:
                return np.asarray(self.loo_likelihood_gradient_batch(thetas)[0], dtype=float)
            except LinAlgError:
                pass  # a trial vector whose matrix is not positive definite: score them one by one (-1e50 for that one)
        with catch_warnings():
            simplefilter("ignore")
            return array([float(self.model_selector(t)) for t in thetas])

    def bfgs_cost_func(self, theta: ndarray):
        y, grad_y = self.model_selector_gradient(theta)
        return -y, -grad_y

    def prepare_gradient(self):
        """(extension) Allocate the likelihood gradient's device workspaces now rather than inside the first
        `marginal_likelihood_gradient` call (gpmi_prepare_gradient): at N = 16384 that first call otherwise costs twice a
        later one.  Called by the gradient-based hyper-parameter search before its first evaluation; a no-op for covariance
        functions that run through the dense path."""
        if not self._generic and self._mix is None and self.n_points > 4096:
            n_theta = np.arange(self.n_hyperpars)[self.cov_slice][self._stat_slice].size  # stationary-kernel parameters
            self.engine.prepare_gradient(n_theta)

    def launch_bfgs(self, x0: ndarray):
        return fmin_l_bfgs_b(func=self.bfgs_cost_func, x0=x0, approx_grad=False, bounds=self.hp_bounds)

    def multistart_bfgs(self, starts: int = None, n_processes: int = 1):
        if starts is None:
            starts = int(2 * sqrt(len(self.hp_bounds))) + 1
        lwr, upr = [array([k[i] for k in self.hp_bounds]) for i in [0, 1]]
        # random starts from the legacy global RNG plus the bounds centre (regression.py:589-594)
        starting_positions = [lwr + (upr - lwr) * random(size=len(self.hp_bounds)) for _ in range(starts - 1)]
        starting_positions.append(0.5 * (lwr + upr))
        if n_processes > 1:
            # the reference farms the starts over a multiprocessing.Pool (regression.py:597-601); here
            # `n_processes` counts GPU workers: when this process is one rank of a multi-GPU job (one process per
            # GPU, see inference_amd.sharding) the starts are block-sharded over the ranks and ONE all-gather
            # returns (theta*, f*) of every start - never a fork of a process that holds a device context.
            from inference_amd import sharding

            if sharding.world()[1] > 1:
                thetas, fvals = sharding.multistart_sweep(self, array(starting_positions))
                return thetas[int(np.argsort(fvals, kind="stable")[0])]
        self.prepare_gradient()
        if self._lockstep_search():
            # every start advances in lockstep (gp/_lockstep.py: SciPy's own L-BFGS-B through its reverse-communication
            # interface, iterates identical to fmin_l_bfgs_b's): one batched device evaluation of the objective and
            # its gradient per round instead of one latency-bound call per start and iteration
            from ._lockstep import lockstep_lbfgsb

            batch = (self.loo_likelihood_gradient_batch if self.model_selector_gradient == self.loo_likelihood_gradient
                     else self.marginal_likelihood_gradient_batch)

            def neg_batch(X):
                f, g = batch(X)
                return -f, -g

            # a round in which ONE start is still running must evaluate it with the lockstep kernels too (a batch of one
            # otherwise takes the single-evaluation kernels, whose gradient differs in the last bits - same LML, gradient
            # within 1e-11): with this a start's iterates do not depend on which other starts share its rounds
            # (tests: test_lockstep_search_is_independent_of_the_grouping)
            was = getattr(self, "_batch_independent", False)
            self.batch_independent_values(True)
            try:
                results = lockstep_lbfgsb(neg_batch, array(starting_positions), self.hp_bounds)
            finally:
                self.batch_independent_values(was)
        else:
            results = [self.launch_bfgs(x0) for x0 in starting_positions]
        # (start, optimum, objective) of every run, in the order of the starts: what the search did, for inspection
        self.search_log = [(array(x0), array(r[0]), float(r[1])) for x0, r in zip(starting_positions, results)]
        return sorted(results, key=lambda r: r[1])[0][0]

    def _lockstep_search(self):
        """The multi-start search runs in lockstep when its objective is the marginal likelihood or (round 4) the
        leave-one-out likelihood of a kernel with a fused device gradient and the problem is small enough for batched
        (lockstep) device evaluations; the values of a start are then those of `launch_bfgs` evaluated through the same
        batched kernels (`batch_independent_values`).  HeteroscedasticNoise - one variance per point and evaluation - rides
        along (gpmi_lml_grad_batch_noise; since round 5 also for the leave-one-out objective: gpmi_loo_grad_batch_noise), and
        so do ChangePoint mixtures (window weights per point and evaluation: gpmi_lml_grad_batch_mix, since round 5
        gpmi_loo_grad_batch_mix and any number of regions: the row-sum weights of `_mix_window_terms`).  Mixtures with
        per-point noise on top have no batched kernels: those starts run one after another."""
        lml = self.model_selector_gradient == self.marginal_likelihood_gradient
        loo = self.model_selector_gradient == self.loo_likelihood_gradient
        mix_ok = self._mix is None or self._het_slice is None
        return ((lml or loo) and not self._generic and mix_ok
                and self._y_cov is None and self.engine.capacity() <= 4096)

    def __str__(self):
        pad = max(len(label) for label in self.hyperpar_labels) + 2
        lines = ["\n[ GpRegressor hyperparameters ]\n"]
        for label, val in zip(self.hyperpar_labels, self.hyperpars):
            lines.append(f"{label:>{pad}} = {val:.4}\n")
        return "".join(lines)
