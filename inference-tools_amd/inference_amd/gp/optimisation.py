"""
`GpOptimiser` — drop-in for `inference.gp.GpOptimiser` (reference:
inference/gp/optimisation.py:14-292): Bayesian optimisation driver which owns a
device-backed `GpRegressor`, proposes the next evaluation by maximising an
acquisition function (multi-start L-BFGS-B or differential evolution on the host,
every objective evaluation on the device) and re-fits when an evaluation is added.
`plot_results` (matplotlib) is out of scope.
"""
from collections.abc import Sequence
from inspect import isclass

from numpy import append, array, ndarray
from scipy.optimize import differential_evolution, fmin_l_bfgs_b

from inference_amd.gp.acquisition import AcquisitionFunction, ExpectedImprovement
from inference_amd.gp.covariance import CovarianceFunction, SquaredExponential
from inference_amd.gp.mean import ConstantMean, MeanFunction
from inference_amd.gp.regression import GpRegressor


class GpOptimiser:
    """
    :param x: coordinates of the evaluations made so far, (N, d) or (N,) for d = 1.
    :param y: objective values at those coordinates.
    :param bounds: iterable of (lower, upper) pairs, one per dimension.
    :param y_err: optional standard errors of `y`.
    :param hyperpars: optional fixed hyper-parameters for the first fit.
    :param kernel, mean, cross_val, optimizer, n_processes: forwarded to `GpRegressor`.
    :param acquisition: acquisition class or instance (default `ExpectedImprovement`).
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
    ):
        self.x = x if isinstance(x, ndarray) else array(x)
        if self.x.ndim == 1:
            self.x = self.x.reshape([self.x.size, 1])
        self.y = y if isinstance(y, ndarray) else array(y)
        self.y_err = y_err if isinstance(y_err, (ndarray, type(None))) else array(y_err)

        self.bounds = bounds
        self.kernel = kernel
        self.mean = mean
        self.cross_val = cross_val
        self.n_processes = n_processes
        self.optimizer = optimizer

        self.gp = GpRegressor(
            x=x,
            y=y,
            y_err=y_err,
            hyperpars=hyperpars,
            kernel=kernel,
            mean=mean,
            cross_val=cross_val,
            optimizer=self.optimizer,
            n_processes=self.n_processes,
        )

        self.acquisition = acquisition() if isclass(acquisition) else acquisition
        self.acquisition.update_gp(self.gp)

        self.acquisition_max_history = []
        self.convergence_metric_history = []
        self.iteration_history = []

    def __call__(self, x):
        return self.gp(x)

    def add_evaluation(self, new_x: ndarray, new_y: ndarray, new_y_err: ndarray = None):
        """Append an evaluation and re-fit from scratch, hyper-parameter search included
        (optimisation.py:136-190)."""
        new_x = new_x if isinstance(new_x, ndarray) else array(new_x)
        if new_x.shape != (1, self.x.shape[1]):
            new_x = new_x.reshape((1, self.x.shape[1]))
        new_y = new_y if isinstance(new_y, ndarray) else array(new_y)
        good_type = isinstance(new_y_err, (ndarray, type(None)))
        new_y_err = new_y_err if good_type else array(new_y_err)

        self.acquisition_max_history.append(self.acquisition(new_x))
        self.convergence_metric_history.append(self.acquisition.convergence_metric(new_x))
        self.iteration_history.append(self.y.size + 1)

        self.x = append(self.x, new_x, axis=0)
        self.y = append(self.y, new_y)

        if self.y_err is not None:
            if new_y_err is not None:
                self.y_err = append(self.y_err, new_y_err)
            else:
                raise ValueError(
                    """\n
                    \r[ GpOptimiser error ]
                    \r>> 'new_y_err' argument of the 'add_evaluation' method must be
                    \r>> specified if the 'y_err' argument was specified when the
                    \r>> instance of GpOptimiser was initialised.
                    """
                )

        self.gp = GpRegressor(
            x=self.x,
            y=self.y,
            y_err=self.y_err,
            kernel=self.kernel,
            mean=self.mean,
            cross_val=self.cross_val,
            optimizer=self.optimizer,
            n_processes=self.n_processes,
        )
        self.mu_max = self.y.max()
        self.acquisition.update_gp(self.gp)

    def diff_evo(self):
        opt_result = differential_evolution(self.acquisition.opt_func, self.bounds, popsize=30)
        funcval = opt_result.fun
        if hasattr(funcval, "__len__"):
            funcval = funcval[0]
        return opt_result.x, funcval

    def launch_bfgs(self, x0: ndarray):
        return fmin_l_bfgs_b(
            self.acquisition.opt_func_gradient, x0, approx_grad=False, bounds=self.bounds, pgtol=1e-10
        )

    def multistart_bfgs(self):
        starting_positions = self.acquisition.starting_positions(self.bounds)
        results = [self.launch_bfgs(x0) for x0 in starting_positions]
        best = sorted(results, key=lambda r: float(r[1]))[0]
        return best[0], float(best[1])

    def propose_evaluation(self, optimizer=None):
        """Location of the next evaluation: the maximiser of the acquisition function
        (optimisation.py:225-249)."""
        opt = optimizer if optimizer is not None else self.optimizer
        if opt == "bfgs":
            proposed_ev, max_acq = self.multistart_bfgs()
        else:
            proposed_ev, max_acq = self.diff_evo()
        if hasattr(proposed_ev, "__len__") and len(proposed_ev) == 1:
            proposed_ev = proposed_ev[0]
        return proposed_ev
