"""
`GpLinearInverter` — drop-in for `inference.gp.GpLinearInverter` (reference:
inference/gp/inversion.py:11-249): Gaussian-process linear inversion.  With a linear forward model
y = A f + noise and a GP prior f ~ N(mu(theta), K(theta)) over the model parameters, the evidence and the
posterior follow from one m x m factorisation, J = A K A^T + Sigma = L L^T; everything O(m n^2 + m^3) runs
on the device (gpmi_linv_*, include/gpmi.h) with the kernels of the regression path, the Nelder-Mead search
over theta stays on the host as in the reference.

Constructor arguments, attributes (`A, y, cov, mean, n_hyperpars, mean_slice, cov_slice,
hyperpar_labels`) and methods are those of the reference.  The posterior is evaluated in its Woodbury
form, K - K A^T J^-1 A K (symmetric positive-definite solves only), which equals the reference's
solve(I + K A^T Sigma^-1 A, K) (inversion.py:150-155).  Supported priors: `SquaredExponential`,
`RationalQuadratic`, each optionally `+ WhiteNoise()`; anything else raises `NotImplementedError`
instead of silently running on the host.
"""
from inspect import isclass

import numpy as np
from numpy import ndarray
from numpy.linalg import LinAlgError
from scipy.optimize import minimize

from inference_amd._engine import LinvEngine
from inference_amd.gp import _messages as msg
from inference_amd.gp.covariance import CovarianceFunction, SquaredExponential, device_plan, heteroscedastic_slice
from inference_amd.gp.mean import ConstantMean, MeanFunction


def _instance(obj):
    return obj() if isclass(obj) else obj


class GpLinearInverter:
    """
    :param y: the data, 1-D.
    :param y_err: standard errors of the data (diagonal likelihood covariance), 1-D.
    :param model_matrix: linear forward model, shape (len(y), number of model parameters).
    :param parameter_spatial_positions: positions of the model parameters, shape (parameters, dimensions).
    :param prior_covariance_function: covariance class or instance (default `SquaredExponential`).
    :param prior_mean_function: mean class or instance (default `ConstantMean`).
    """

    def __init__(
        self,
        y: ndarray,
        y_err: ndarray,
        model_matrix: ndarray,
        parameter_spatial_positions: ndarray,
        prior_covariance_function: CovarianceFunction = SquaredExponential,
        prior_mean_function: MeanFunction = ConstantMean,
        device=None,
    ):
        msg.check_inverter_shapes(y, y_err, model_matrix, parameter_spatial_positions)

        self.A = model_matrix
        self.y = y
        self.y_err = y_err
        self.positions = parameter_spatial_positions

        self.cov = _instance(prior_covariance_function)
        self.cov.pass_spatial_data(parameter_spatial_positions)
        if self.cov.bounds is None:
            self.cov.bounds = [(None, None)] * self.cov.n_params
        self.mean = _instance(prior_mean_function)
        self.mean.pass_spatial_data(parameter_spatial_positions)
        if self.mean.bounds is None:
            self.mean.bounds = [(None, None)] * self.mean.n_params

        self.n_hyperpars = self.mean.n_params + self.cov.n_params
        self.mean_slice = slice(0, self.mean.n_params)
        self.cov_slice = slice(self.mean.n_params, self.n_hyperpars)
        self.hyperpar_labels = [*self.mean.hyperpar_labels, *self.cov.hyperpar_labels]

        # SquaredExponential / RationalQuadratic (+ WhiteNoise) priors are built on the device; for any other
        # CovarianceFunction object (the reference takes any, inversion.py:117-127) the host evaluates the object's
        # own build_covariance / covariance_and_gradients and the device does every O(n^3) step (gpmi_linv_*_dense)
        plan = device_plan(self.cov)
        self._dense = plan is None or plan[0] < 0 or heteroscedastic_slice(self.cov) is not None  # -1: ChangePoint mixture
        if not self._dense:
            self._kernel_id, self._stat, self._stat_slice, self._wn_index = plan
        self._device = device
        self._engine = None

    # ---------------------------------------------------------------------------------
    @property
    def engine(self) -> LinvEngine:
        if self._engine is None:
            self._engine = LinvEngine(self.positions, self.A, self.y, self.y_err, device=self._device)
        return self._engine

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engine"] = None  # device state is rebuilt on first use
        return state

    def _device_args(self, theta):
        theta = np.asarray(theta, dtype=float)
        theta_cov = theta[self.cov_slice]
        extra = float(np.exp(2 * theta_cov[self._wn_index])) if self._wn_index is not None else 0.0
        return np.ascontiguousarray(theta_cov[self._stat_slice]), extra

    @staticmethod
    def _check(info):
        if info != 0:  # numpy.linalg.cholesky / scipy.linalg.solve of the reference raise LinAlgError
            raise LinAlgError("Matrix is not positive definite")

    # ---------------------------------------------------------------------------------
    def calculate_posterior(self, theta: ndarray):
        """Posterior mean and covariance of the model parameters (inversion.py:138-156)."""
        theta = np.asarray(theta, dtype=float)
        prior_mean = self.mean.build_mean(theta[self.mean_slice])
        if self._dense:
            mean, cov, info = self.engine.posterior_dense(self.cov.build_covariance(theta[self.cov_slice]), prior_mean)
        else:
            theta_stat, extra = self._device_args(theta)
            mean, cov, info = self.engine.posterior(self._kernel_id, theta_stat, extra, prior_mean, with_cov=True)
        self._check(info)
        return mean, cov

    def calculate_posterior_mean(self, theta: ndarray) -> ndarray:
        """Posterior mean only (inversion.py:158-175)."""
        theta = np.asarray(theta, dtype=float)
        prior_mean = self.mean.build_mean(theta[self.mean_slice])
        if self._dense:
            mean, _, info = self.engine.posterior_dense(self.cov.build_covariance(theta[self.cov_slice]), prior_mean,
                                                        with_cov=False)
        else:
            theta_stat, extra = self._device_args(theta)
            mean, _, info = self.engine.posterior(self._kernel_id, theta_stat, extra, prior_mean, with_cov=False)
        self._check(info)
        return mean

    def marginal_likelihood(self, theta: ndarray) -> float:
        """Log-marginal likelihood without the 2 pi constant (inversion.py:177-191)."""
        theta = np.asarray(theta, dtype=float)
        prior_mean = self.mean.build_mean(theta[self.mean_slice])
        if self._dense:
            lml, info = self.engine.lml_dense(self.cov.build_covariance(theta[self.cov_slice]), prior_mean)
        else:
            theta_stat, extra = self._device_args(theta)
            lml, info = self.engine.lml(self._kernel_id, theta_stat, extra, prior_mean)
        self._check(info)
        return float(lml)

    def marginal_likelihood_gradient(self, theta: ndarray):
        """Log-marginal likelihood and its gradient (inversion.py:193-217)."""
        theta = np.asarray(theta, dtype=float)
        mu, grad_mu = self.mean.mean_and_gradients(theta[self.mean_slice])
        if self._dense:
            K, grad_K = self.cov.covariance_and_gradients(theta[self.cov_slice])
            lml, G, w, info = self.engine.lml_grad_dense(K, mu)
            self._check(info)
            grad = np.zeros(self.n_hyperpars)
            grad[self.mean_slice] = np.array([(w * dmu).sum() for dmu in grad_mu])
            # 1/2 sum (alpha alpha^T - J^-1) o (A dK A^T)^T = 1/2 sum (w w^T - A^T J^-1 A) o dK^T, w = A^T alpha
            Q = w[:, None] * w[None, :] - G
            grad[self.cov_slice] = np.array([0.5 * (Q * dK.T).sum() for dK in grad_K])
            return float(lml), grad
        theta_stat, extra = self._device_args(theta)
        lml, g_stat, trace_q, at_alpha, info = self.engine.lml_grad(self._kernel_id, theta_stat, extra, mu)
        self._check(info)
        grad = np.zeros(self.n_hyperpars)
        # sum_i alpha_i (A dmu)_i = (A^T alpha) . dmu   (inversion.py:211-212)
        grad[self.mean_slice] = np.array([(at_alpha * dmu).sum() for dmu in grad_mu])
        g_cov = np.zeros(self.cov.n_params)
        g_cov[self._stat_slice] = g_stat
        if self._wn_index is not None:
            g_cov[self._wn_index] = extra * trace_q  # 1/2 sum Q o (2 sigma^2 I), covariance.py:171-175
        grad[self.cov_slice] = g_cov
        return float(lml), grad

    def optimize_hyperparameters(self, initial_guess: ndarray) -> ndarray:
        """Nelder-Mead maximisation of the marginal likelihood (inversion.py:219-249)."""
        if initial_guess.size != self.n_hyperpars:
            raise ValueError(msg.framed_plain(
                "GpLinearInverter", f"There are a total of {self.n_hyperpars} hyper-parameters,",
                f"but {initial_guess.size} values were given in 'initial_guess'."))
        bounds = [*self.mean.bounds, *self.cov.bounds]
        found = minimize(fun=lambda t: -self.marginal_likelihood(t), x0=initial_guess, method="Nelder-Mead",
                         bounds=bounds)
        return found.x
