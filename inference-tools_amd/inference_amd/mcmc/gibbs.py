"""
Gibbs-sampling Markov chain — the caller of the GP log-marginal-likelihood in
BASELINE config 5.  Mirrors the step semantics of the reference's `GibbsChain`
(inference/mcmc/gibbs.py:593-656) and of its per-parameter proposal / width
adaptation (`Parameter`, gibbs.py:16-160): every step is a sequence of 1-D
Metropolis-Hastings updates, one per parameter, each retried until accepted,
with proposal widths tuned towards a 50 % acceptance rate.

Only what `ParallelTempering` needs is provided (`inv_temp`, `take_step`,
`get_last`, `replace_last`, `probs`, sample access, boundaries); plotting, KDE
marginals and save / load of the reference are out of scope.

MI355X-specific addition: `advance_lockstep` advances MANY chains together so
that each round of proposals is ONE batched posterior evaluation on the device
(`GpRegressor.marginal_likelihood_batch`).  Chains keep their own random
generators (one per chain, one per parameter, as in the reference), so lockstep
and one-by-one execution produce identical trajectories.
"""
from copy import copy

import numpy as np
from numpy import array, exp, float64, isfinite, log, sqrt
from numpy.random import default_rng


class Parameter:
    """Proposal state of one model parameter (reference: gibbs.py:16-160)."""

    def __init__(self, value: float, sigma: float):
        self.samples = [value]
        self.sigma = sigma
        self.rng = default_rng()
        # acceptance statistics since the last width assessment
        self.avg = 0
        self.var = 0
        self.num = 0
        self.sigma_values = [copy(self.sigma)]
        self.sigma_checks = [0.0]
        self.try_count = 0
        # adaptation settings (gibbs.py:41-46)
        self.target_rate = 0.25
        self.max_tries = 50
        self.chk_int = 100
        self.growth_factor = 1.75
        self.adjust_rate = 0.25
        self.non_negative = False
        self.bounded = False
        self.lower = 0.0
        self.upper = 0.0
        self.width = 0.0

    def set_boundaries(self, lower, upper):
        if not lower < upper:
            raise ValueError("Upper limit must be greater than lower limit")
        self.lower, self.upper, self.width = lower, upper, upper - lower
        self.bounded = True

    def remove_boundaries(self):
        self.bounded = False
        self.lower = self.upper = self.width = 0.0

    def _fold_into_bounds(self, value):
        """Reflect `value` back into [lower, upper] as many times as needed (a triangle wave of period 2 width)."""
        offset = value - self.lower
        inside = offset % self.width
        bounces = offset // self.width
        return self.lower + inside if bounces % 2 == 0 else self.upper - inside

    def proposal(self):
        """Draw the next 1-D proposal (behaviour of gibbs.py:88-122): a normal step from the last sample; a
        parameter that has already failed `max_tries` times in this update has its width quartered first;
        bounded parameters are reflected inside, non-negative ones mirrored at zero."""
        self.try_count += 1
        if self.try_count > self.max_tries:
            self.adjust_sigma(0.25)
        step = self.rng.normal(loc=self.samples[-1], scale=self.sigma)
        if self.bounded:
            return self._fold_into_bounds(step)
        return abs(step) if self.non_negative else step

    def submit_accept_prob(self, p: float):
        """Record the acceptance probability of one proposal; every `chk_int` records the width is reviewed."""
        self.num += 1
        self.avg += p
        self.var += p * (1 - p)
        if self.num >= self.chk_int:
            self.update_epsilon()

    def update_epsilon(self):
        """Width review (behaviour of gibbs.py:132-148).  The number of acceptances is approximately normal
        (Poisson-binomial); the width changes only when the target rate lies outside two standard deviations
        of the observed mean rate, otherwise the review interval grows."""
        rate = self.avg / self.num
        spread = sqrt(self.var) / self.num
        if rate - 2 * spread < self.target_rate < rate + 2 * spread:
            self.chk_int = int((self.growth_factor * self.chk_int) * 0.1) * 10
            return
        factor = (log(self.target_rate) / log(rate)) ** self.adjust_rate
        self.adjust_sigma(min(max(factor, 0.1), 3.0))

    def adjust_sigma(self, ratio: float):
        """Scale the proposal width, log it and restart the acceptance statistics."""
        self.sigma *= ratio
        self.sigma_values.append(copy(self.sigma))
        self.sigma_checks.append(len(self.samples))
        self.avg = self.var = self.num = 0

    def add_sample(self, s):
        self.samples.append(s)
        self.try_count = 0


class GibbsChain:
    """
    :param posterior: callable `theta (ndarray) -> float` log-probability
        (e.g. `GpRegressor.marginal_likelihood`).
    :param start: starting parameter vector.
    :param widths: initial proposal widths (default 5 % of `start`, 1.0 where it is zero).
    :param temperature: chain temperature T; the chain samples posterior ** (1 / T).
    """

    def __init__(self, posterior, start, widths=None, temperature: float = 1.0, display_progress: bool = True):
        self.inv_temp = 1.0 / temperature
        self.rng = default_rng()
        self.posterior = posterior
        self._validate_posterior(posterior, start)
        if widths is None:
            widths = [v * 0.05 if v != 0 else 1.0 for v in start]
        self.params = [Parameter(value=v, sigma=s) for v, s in zip(start, widths)]
        for p in self.params:
            p.target_rate = 0.5  # optimal for 1-D updates (gibbs.py:621-625)
        self.chain_length = 1
        self.n_parameters = len(start)
        self.probs = [self.posterior(self.get_last()) * self.inv_temp]
        self.display_progress = display_progress

    def _validate_posterior(self, posterior, start):
        """base.py:266-296."""
        name = self.__class__.__name__
        if not callable(posterior):
            raise ValueError(f"\n[ {name} error ]\n>> The given 'posterior' is not a callable object.")
        prob = posterior(start)
        if not isinstance(prob, float):
            raise ValueError(
                f"\n[ {name} error ]\n>> The given 'posterior' must return a float or a type which derives "
                f"from float, however the returned value has type:\n>> {type(prob)}"
            )
        if not isfinite(prob):
            raise ValueError(
                f"\n[ {name} error ]\n>> The given 'posterior' must return a finite value for the given "
                f"'start' parameter values, but instead returns a value of:\n>> {prob}"
            )

    # -- stepping ----------------------------------------------------------------------
    def take_step(self):
        """One Gibbs step: a retried-until-accepted 1-D MH update per parameter (gibbs.py:627-656)."""
        p_old = self.probs[-1]
        prop = self.get_last()
        p_new = p_old
        for i, par in enumerate(self.params):
            while True:
                prop[i] = par.proposal()
                p_new = self.posterior(prop) * self.inv_temp
                if self._mh_test(par, p_new, p_old):
                    break
            p_old = p_new
        self._commit(prop, p_new)

    def _mh_test(self, par, p_new, p_old) -> bool:
        if p_new > p_old:
            par.submit_accept_prob(1.0)
            return True
        acceptance_prob = exp(p_new - p_old)
        par.submit_accept_prob(acceptance_prob)
        return bool(self.rng.random() < acceptance_prob)

    def _commit(self, prop, p_new):
        for v, par in zip(prop, self.params):
            par.add_sample(v)
        self.probs.append(p_new)
        self.chain_length += 1

    def advance(self, m: int):
        for _ in range(m):
            self.take_step()

    # -- state access -------------------------------------------------------------------
    def get_last(self):
        return array([p.samples[-1] for p in self.params], dtype=float64)

    def replace_last(self, theta):
        for p, t in zip(self.params, theta):
            p.samples[-1] = t

    def get_parameter(self, index: int, burn: int = 1, thin: int = 1):
        return array(self.params[index].samples[burn::thin])

    def get_probabilities(self, burn: int = 1, thin: int = 1):
        return array(self.probs[burn::thin])

    def get_sample(self, burn: int = 1, thin: int = 1):
        return array([self.get_parameter(i, burn=burn, thin=thin) for i in range(self.n_parameters)]).T

    def set_non_negative(self, parameter: int, flag=True):
        self.params[parameter].non_negative = flag

    def set_boundaries(self, parameter: int, boundaries, remove=False):
        if remove:
            self.params[parameter].remove_boundaries()
        else:
            self.params[parameter].set_boundaries(*boundaries)


def advance_lockstep(chains, n: int, batch_posterior):
    """Advance every chain by `n` Gibbs steps with batched posterior evaluations.

    `batch_posterior(thetas (B, P)) -> (B,)` must agree element-wise with each chain's own
    `posterior`.  Every round, every chain that still has work proposes a value for ITS current
    parameter and all proposals are evaluated in one call; a chain whose proposal is accepted moves
    on to its next parameter (and its next step) at once, one whose proposal is rejected retries
    (the retry-until-accept loop of gibbs.py:635-648).  Chains therefore drift apart inside a call
    and only meet again at its end - a chain draws from its own generators only and never sees the
    others, so its trajectory is the one it would follow alone - and the batches stay full until the
    first chains finish, instead of shrinking 64 -> 1 for every parameter.
    Returns the number of posterior evaluations made."""
    if not chains or n <= 0:
        return 0
    P = chains[0].n_parameters
    evals = 0
    step = [0] * len(chains)  # completed steps
    par = [0] * len(chains)   # parameter being updated
    p_old = [c.probs[-1] for c in chains]
    p_acc = list(p_old)
    prop = [c.get_last() for c in chains]
    active = list(range(len(chains)))
    while active:
        for c in active:
            prop[c][par[c]] = chains[c].params[par[c]].proposal()
        vals = batch_posterior(array([prop[c] for c in active]))
        evals += len(active)
        still = []
        for c, v in zip(active, vals):
            chain = chains[c]
            p_new = float(v) * chain.inv_temp
            if chain._mh_test(chain.params[par[c]], p_new, p_old[c]):
                p_old[c] = p_new
                p_acc[c] = p_new
                par[c] += 1
                if par[c] == P:
                    chain._commit(prop[c], p_acc[c])
                    step[c] += 1
                    par[c] = 0
                    if step[c] < n:
                        p_old[c] = chain.probs[-1]
                        p_acc[c] = p_old[c]
                        prop[c] = chain.get_last()
            if step[c] < n:
                still.append(c)
        active = still
    return evals
