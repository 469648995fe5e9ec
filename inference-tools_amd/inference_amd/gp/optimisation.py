"""
`GpOptimiser` — drop-in for `inference.gp.GpOptimiser` (reference:
inference/gp/optimisation.py:14-292): Bayesian optimisation driver which owns a
device-backed `GpRegressor`, proposes the next evaluation by maximising an
acquisition function (multi-start L-BFGS-B or differential evolution on the host,
every objective evaluation on the device) and re-fits when an evaluation is added.
`plot_results` (matplotlib) is out of scope.

Same constructor arguments, public attributes (`x, y, y_err, bounds, gp, acquisition,
acquisition_max_history, convergence_metric_history, iteration_history`) and methods as the
reference; the regressor is (re)built in one place, `_fit_gp`, so every re-fit goes through the
same device path: one K-build + blocked Cholesky per hyper-parameter evaluation.
"""
from collections.abc import Sequence
from inspect import isclass

import numpy as np
from numpy import ndarray
from scipy.optimize import differential_evolution, fmin_l_bfgs_b

from inference_amd.gp import _messages as msg
from inference_amd.gp.acquisition import AcquisitionFunction, ExpectedImprovement
from inference_amd.gp.covariance import CovarianceFunction, SquaredExponential
from inference_amd.gp.mean import ConstantMean, MeanFunction
from inference_amd.gp.regression import GpRegressor


def _optional_array(value):
    return None if value is None else np.asarray(value)


class GpOptimiser:
    """
    :param x: coordinates of the evaluations made so far, (N, d) or (N,) for d = 1.
    :param y: objective values at those coordinates.
    :param bounds: iterable of (lower, upper) pairs, one per dimension.
    :param y_err: optional standard errors of `y`.
    :param hyperpars: optional fixed hyper-parameters for the first fit.
    :param kernel, mean, cross_val, optimizer, n_processes: forwarded to `GpRegressor`.
    :param acquisition: acquisition class or instance (default `ExpectedImprovement`).
    :param reuse_hyperpars: (extension, default False = the reference's behaviour) keep the hyper-parameters of
        the first fit when evaluations are added: `add_evaluation` then appends the point to the fitted model
        in O(N^2) (`GpRegressor.add_point`) instead of searching the hyper-parameters again and re-factorising
        (O(N^3) per likelihood evaluation, optimisation.py:177-186).
    :param device: (extension) index of the GPU (one process per GPU: the rank's local device).
    """

    def __init__(
        self,
        x: ndarray,
        y: ndarray,
        bounds: Sequence,
        y_err: ndarray = None,
        hyperpars: ndarray = None,
        kernel: CovarianceFunction = SquaredExponential,
        mean: MeanFunction = ConstantMean,
        cross_val: bool = False,
        acquisition: AcquisitionFunction = ExpectedImprovement,
        optimizer: str = "bfgs",
        n_processes: int = 1,
        reuse_hyperpars: bool = False,
        device: int = 0,
    ):
        coords = np.asarray(x)
        self.device = int(device)  # (extension) the GPU every regressor of this optimiser lives on
        self.x = coords.reshape([coords.size, 1]) if coords.ndim == 1 else coords
        self.y = np.asarray(y)
        self.y_err = _optional_array(y_err)
        self.bounds = bounds

        # everything a re-fit needs (optimisation.py:117-127, 178-188)
        self.kernel, self.mean = kernel, mean
        self.cross_val = cross_val
        self.optimizer = optimizer
        self.n_processes = n_processes
        self.reuse_hyperpars = bool(reuse_hyperpars)

        self.acquisition = acquisition() if isclass(acquisition) else acquisition
        self.acquisition_max_history = []
        self.convergence_metric_history = []
        self.iteration_history = []
        self._fit_gp(hyperpars)

    def _fit_gp(self, hyperpars=None):
        """(Re)build the regressor on the current evaluations and point the acquisition at it."""
        self.gp = GpRegressor(
            self.x,
            self.y,
            y_err=self.y_err,
            hyperpars=hyperpars,
            kernel=self.kernel,
            mean=self.mean,
            cross_val=self.cross_val,
            optimizer=self.optimizer,
            n_processes=self.n_processes,
            reserve=256 if self.reuse_hyperpars else 0,
            device=self.device,
        )
        self.acquisition.update_gp(self.gp)

    def __call__(self, x):
        return self.gp(x)

    def add_evaluation(self, new_x: ndarray, new_y: ndarray, new_y_err: ndarray = None):
        """Append an evaluation and re-fit from scratch, hyper-parameter search included
        (optimisation.py:136-190)."""
        point = np.asarray(new_x).reshape((1, self.x.shape[1]))
        value = np.asarray(new_y)
        error = _optional_array(new_y_err)

        if self.y_err is not None and error is None:
            raise ValueError(msg.NEW_Y_ERR_REQUIRED)
        # how promising this point looked under the *previous* model: evaluated now, recorded only once the model has
        # accepted the point - a failed factor update (a duplicate proposal at fixed hyper-parameters: pivot <= 0,
        # LinAlgError) leaves x / y AND the three histories as they were
        acq_value = self.acquisition(point)
        metric = self.acquisition.convergence_metric(point)
        if self.reuse_hyperpars:
            self.gp.add_point(point, value, error)
        self.acquisition_max_history.append(acq_value)
        self.convergence_metric_history.append(metric)
        self.iteration_history.append(self.y.size + 1)
        self.x = np.append(self.x, point, axis=0)
        self.y = np.append(self.y, value)
        if self.y_err is not None:
            self.y_err = np.append(self.y_err, error)

        if self.reuse_hyperpars:
            self.acquisition.update_gp(self.gp)
        else:
            self._fit_gp()
        self.mu_max = self.y.max()

    # -- acquisition maximisers (optimisation.py:192-223) ------------------------------------
    def diff_evo(self):
        found = differential_evolution(self.acquisition.opt_func, self.bounds, popsize=30)
        return found.x, float(np.ravel(found.fun)[0])

    def launch_bfgs(self, x0: ndarray):
        return fmin_l_bfgs_b(
            self.acquisition.opt_func_gradient, x0, approx_grad=False, bounds=self.bounds, pgtol=1e-10
        )

    def multistart_bfgs(self):
        """One L-BFGS-B run (pgtol 1e-10) per starting position, best result wins (optimisation.py:202-223).  The
        reference runs them one after another (or over a multiprocessing.Pool); here all runs advance in lockstep
        (`_lockstep.lockstep_lbfgsb`): every round evaluates the acquisition function and its gradient for all
        runs still iterating in ONE batched device call, so a proposal at N = 4096 training points costs tens of
        device calls instead of ~10^5 single-point ones.  Acquisition objects without a batched gradient keep
        the serial path."""
        starts = self.acquisition.starting_positions(self.bounds)
        batch = getattr(self.acquisition, "opt_func_gradient_batch", None)
        if batch is None:
            runs = [self.launch_bfgs(x0) for x0 in starts]
        else:
            from inference_amd.gp._lockstep import lockstep_lbfgsb

            runs = lockstep_lbfgsb(batch, np.array(starts), self.bounds, pgtol=1e-10)
        where, lowest, _ = min(runs, key=lambda run: float(run[1]))
        return where, float(lowest)

    def propose_evaluation(self, optimizer=None):
        """Location of the next evaluation: the maximiser of the acquisition function
        (optimisation.py:225-249)."""
        method = self.optimizer if optimizer is None else optimizer
        proposal, _ = self.multistart_bfgs() if method == "bfgs" else self.diff_evo()
        if np.ndim(proposal) > 0 and len(proposal) == 1:
            return proposal[0]
        return proposal
