"""
Acquisition functions for Bayesian optimisation — host-side mirror of
`inference/gp/acquisition.py` (reference): same classes, methods and return
conventions (`__call__`, `opt_func`, `opt_func_gradient`, `starting_positions`,
`update_gp`, `convergence_metric`).

The GP quantities (predictive mean / standard deviation and their spatial
derivatives) come from the device through `GpRegressor`; the O(M) scalar
epilogues (erf / erfcx) stay on the host with the same SciPy special functions as
the reference.  Each class additionally offers `*_batch` methods which evaluate
many candidates with ONE device call — the unit used by `starting_positions`
(20 probes per training point, acquisition.py:26-31) and by config 4 (1000
candidates per step).
"""
import numpy as np
from numpy import array, exp, log, maximum, minimum, ndarray, pi, sqrt
from numpy.random import random
from scipy.special import erf, erfcx


class AcquisitionFunction:
    gp = None
    mu_max: float

    def starting_positions(self, bounds):
        """One L-BFGS start per training point inside the bounds, chosen as the best of 20
        jittered probes (acquisition.py:13-37).  The random numbers are drawn in the reference's
        order; all probes are then ranked from a single batched device evaluation.

        The reference draws `random(size=L)` 20 times per training point inside the bounds and once per point outside,
        point by point; here ONE draw of the same total length is cut up in the same order (the legacy generator fills an
        array sequentially: the numbers, and with them the probes, are the same) - the Python loop over 4096 x 20 probes
        was a third of the wall time of `GpOptimiser.propose_evaluation` at N = 4096 (round 6)."""
        lwr, upr = [array([k[i] for k in bounds], dtype=float) for i in [0, 1]]
        widths = upr - lwr
        lwr += widths * 0.01
        upr -= widths * 0.01
        L = len(widths)
        X = np.asarray(self.gp.x, dtype=float).reshape(len(self.gp.x), L)
        inside = ((X >= lwr) & (X <= upr)).all(axis=1)
        counts = np.where(inside, 20 * L, L)
        offs = np.concatenate(([0], np.cumsum(counts)))
        r = random(size=int(offs[-1]))
        starts = [None] * len(X)
        idx_in = np.flatnonzero(inside)
        for i in np.flatnonzero(~inside):
            starts[i] = lwr + (upr - lwr) * r[offs[i]:offs[i + 1]]
        if len(idx_in):
            take = offs[idx_in][:, None] + np.arange(20 * L)[None, :]
            rr = r[take].reshape(len(idx_in), 20, L)
            samples = X[idx_in][:, None, :] + 0.02 * widths * (2 * rr - 1)
            samples = minimum(upr, maximum(lwr, samples))
            vals = self.opt_func_batch(samples.reshape(-1, L)).reshape(len(idx_in), 20)
            best = np.argsort(vals, axis=1, kind="stable")[:, 0]
            for g, i in enumerate(idx_in):
                starts[i] = samples[g, best[g]]
        return starts

    def update_gp(self, gp):
        self.gp = gp
        self.mu_max = gp.y.max()

    # scalar API in terms of the batched one
    def opt_func(self, x) -> float:
        return float(self.opt_func_batch(self.gp.process_points(x))[0])

    def opt_func_batch(self, points) -> ndarray:
        raise NotImplementedError


class ExpectedImprovement(AcquisitionFunction):
    r"""EI(x) = (z F(z) + P(z)) sigma(x), z = (mu(x) - y_max) / sigma(x) (acquisition.py:44-140)."""

    def __init__(self):
        self.ir2pi = 1 / sqrt(2 * pi)
        self.ir2 = 1.0 / sqrt(2)
        self.rpi2 = sqrt(0.5 * pi)
        self.ln2pi = log(2 * pi)
        self.name = "Expected improvement"
        self.convergence_description = r"$\mathrm{EI}_{\mathrm{max}} \; / \; (y_{\mathrm{max}} - y_{\mathrm{min}})$"

    # -- batched evaluation ---------------------------------------------------------
    def _ln_ei(self, mu, sig):
        """ln EI with the reference's branch at Z = -3 (acquisition.py:88-97)."""
        Z = (mu - self.mu_max) / sig
        out = np.empty_like(Z)
        lo = Z < -3
        zl, sl = Z[lo], sig[lo]
        out[lo] = log(1 + zl * self.cdf_pdf_ratio(zl)) + self.ln_pdf(zl) + log(sl)
        zh, sh = Z[~lo], sig[~lo]
        out[~lo] = log(sh * (zh * self.normal_cdf(zh) + self.normal_pdf(zh)))
        return out, Z, lo

    def call_batch(self, points) -> ndarray:
        mu, sig = self.gp(points)
        ln_ei, Z, lo = self._ln_ei(mu, sig)
        out = np.empty_like(mu)
        out[lo] = exp(ln_ei[lo])  # acquisition.py:80-81
        zh, sh = Z[~lo], sig[~lo]
        out[~lo] = sh * (zh * self.normal_cdf(zh) + self.normal_pdf(zh))  # acquisition.py:83-85
        return out

    def opt_func_batch(self, points) -> ndarray:
        mu, sig = self.gp(points)
        return -self._ln_ei(mu, sig)[0]

    def opt_func_gradient_batch(self, points):
        """(-ln EI, -grad ln EI) for M points: (M,), (M, d) (acquisition.py:99-125)."""
        p = self.gp.process_points(points)
        mu, sig = self.gp(p)
        dmu, dvar = self.gp.spatial_derivatives_batch(p)
        ln_ei, Z, lo = self._ln_ei(mu, sig)
        grad = np.empty_like(dmu)
        if lo.any():
            R = self.cdf_pdf_ratio(Z[lo])
            H = 1 + Z[lo] * R
            grad[lo] = (0.5 * dvar[lo] / sig[lo, None] + R[:, None] * dmu[lo]) / (H * sig[lo])[:, None]
        hi = ~lo
        if hi.any():
            pdf, cdf = self.normal_pdf(Z[hi]), self.normal_cdf(Z[hi])
            EI = sig[hi] * (Z[hi] * cdf + pdf)
            grad[hi] = (0.5 * pdf[:, None] * dvar[hi] / sig[hi, None] + dmu[hi] * cdf[:, None]) / EI[:, None]
        return -ln_ei, -grad

    # -- reference-compatible scalar API --------------------------------------------------
    def __call__(self, x) -> float:
        return self.call_batch(self.gp.process_points(x))[0]

    def opt_func_gradient(self, x):
        val, grad = self.opt_func_gradient_batch(x)
        return array(val[0]), array(grad[0]).squeeze()

    def normal_pdf(self, z):
        return exp(-0.5 * z**2) * self.ir2pi

    def normal_cdf(self, z):
        return 0.5 * (1.0 + erf(z * self.ir2))

    def cdf_pdf_ratio(self, z):
        return self.rpi2 * erfcx(-z * self.ir2)

    def ln_pdf(self, z):
        return -0.5 * (z**2 + self.ln2pi)

    def convergence_metric(self, x):
        return self.__call__(x) / (self.mu_max - self.gp.y.min())


class UpperConfidenceBound(AcquisitionFunction):
    r"""UCB(x) = mu(x) + kappa sigma(x) (acquisition.py:143-192)."""

    def __init__(self, kappa: float = 2.0):
        self.kappa = kappa
        self.name = "Upper confidence bound"
        self.convergence_description = r"$\mathrm{UCB}_{\mathrm{max}} - y_{\mathrm{max}}$"

    def call_batch(self, points) -> ndarray:
        mu, sig = self.gp(points)
        return mu + self.kappa * sig

    def opt_func_batch(self, points) -> ndarray:
        return -self.call_batch(points)

    def opt_func_gradient_batch(self, points):
        p = self.gp.process_points(points)
        mu, sig = self.gp(p)
        dmu, dvar = self.gp.spatial_derivatives_batch(p)
        return -(mu + self.kappa * sig), -(dmu + 0.5 * self.kappa * dvar / sig[:, None])

    def __call__(self, x) -> float:
        return self.call_batch(self.gp.process_points(x))[0]

    def opt_func_gradient(self, x):
        val, grad = self.opt_func_gradient_batch(x)
        return array(val[0]), array(grad[0]).squeeze()

    def convergence_metric(self, x):
        return self.__call__(x) - self.mu_max


class MaxVariance(AcquisitionFunction):
    r"""Pure-learning acquisition: the predictive variance sigma^2(x) (acquisition.py:195-232)."""

    def __init__(self):
        self.name = "Max variance"
        self.convergence_description = r"$\sqrt{\mathrm{Var}\left[x\right]}$"

    def call_batch(self, points) -> ndarray:
        _, sig = self.gp(points)
        return sig**2

    def opt_func_batch(self, points) -> ndarray:
        return -self.call_batch(points)

    def opt_func_gradient_batch(self, points):
        p = self.gp.process_points(points)
        _, sig = self.gp(p)
        _, dvar = self.gp.spatial_derivatives_batch(p)
        return -(sig**2), -dvar

    def __call__(self, x) -> float:
        return self.call_batch(self.gp.process_points(x))[0]

    def opt_func_gradient(self, x):
        val, grad = self.opt_func_gradient_batch(x)
        return array(val).squeeze(), array(grad[0]).squeeze()

    def convergence_metric(self, x):
        return sqrt(self.__call__(x))
