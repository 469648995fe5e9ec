"""
CPU tests of the MCMC callers of the hot path (GibbsChain step semantics, ParallelTempering swap
rule, lockstep batching) against teacher-forced traces of the reference (tests/golden/pt.npz:
every generator seeded by assignment, posterior = the reference GpRegressor.marginal_likelihood).
The posterior used here is the CPU oracle's marginal_likelihood (test infrastructure).
"""
import random

import numpy as np
import pytest
from numpy.random import default_rng

from inference_amd.mcmc import GibbsChain, ParallelTempering, advance_ladders, advance_lockstep
from oracle import gp_oracle as orc


def pt_problem():
    rng = np.random.default_rng(31)
    x = np.sort(rng.uniform(0, 6, 48))
    y = np.sin(x) + 0.3 * np.cos(2.5 * x) + 0.2 * rng.normal(size=48)
    y_err = np.full(48, 0.2)
    start = np.array([y.mean(), np.log(y.std()), np.log(1.0)])
    widths = np.array([0.1, 0.2, 0.2])
    return x, y, y_err, start, widths


def make_chain(posterior, start, widths, bounds, temp, seed):
    ch = GibbsChain(posterior=posterior, start=start, widths=widths, temperature=temp, display_progress=False)
    for i, b in enumerate(bounds):
        ch.set_boundaries(i, b)
    ch.rng = default_rng(seed)
    for i, par in enumerate(ch.params):
        par.rng = default_rng(seed + 1 + i)
    return ch


@pytest.fixture(scope="module")
def problem():
    x, y, e, start, widths = pt_problem()
    gp = orc.OracleGp(x, y, e, kernel=orc.SE)
    return gp, start, widths


def test_gibbs_chain_reproduces_reference_trace(golden, problem):
    g = golden("pt")
    gp, start, widths = problem
    assert np.allclose(np.array(gp.hp_bounds), g["hp_bounds"], rtol=1e-12)
    ch = make_chain(gp.marginal_likelihood, start, widths, g["hp_bounds"], 1.0, 100)
    ch.advance(60)
    assert np.allclose(ch.get_sample(burn=0), g["single_samples"], rtol=0, atol=1e-9)
    assert np.allclose(np.array(ch.probs), g["single_probs"], rtol=1e-10)
    assert np.allclose([p.sigma for p in ch.params], g["single_sigmas"], rtol=1e-12)


def test_parallel_tempering_reproduces_reference_trace(golden, problem):
    g = golden("pt")
    gp, start, widths = problem
    chains = [make_chain(gp.marginal_likelihood, start, widths, g["hp_bounds"], t, 1000 + 10 * k)
              for k, t in enumerate(g["temps"])]
    pt = ParallelTempering(chains)
    assert pt.batch_posterior is None  # plain callable -> sequential stepping
    pt.rng = default_rng(7)
    random.seed(9)
    pt.advance(40, swap_interval=5)
    for k, c in enumerate(pt.return_chains()):
        assert np.allclose(c.get_sample(burn=0), g[f"pt_samples_{k}"], rtol=0, atol=1e-9), k
        assert np.allclose(np.array(c.probs), g[f"pt_probs_{k}"], rtol=1e-10), k
    assert np.array_equal(pt.successful_swaps, g["pt_successful"])
    assert np.array_equal(pt.attempted_swaps, g["pt_attempted"])
    pt.shutdown()


def test_lockstep_equals_sequential(golden, problem):
    """Batched (lockstep) advancement is bit-identical to chain-by-chain stepping, ragged retries included."""
    g = golden("pt")
    gp, start, widths = problem

    def batch(thetas):
        return np.array([gp.marginal_likelihood(t) for t in thetas])

    def ladders():
        out = []
        for lad in range(2):
            chains = [make_chain(gp.marginal_likelihood, start, widths, g["hp_bounds"], t, 5000 + 100 * lad + 10 * k)
                      for k, t in enumerate([1.0, 3.0, 9.0])]
            pt = ParallelTempering(chains, batch_posterior=None)
            pt.rng = default_rng(70 + lad)
            out.append(pt)
        return out

    seq = ladders()
    random.seed(5)
    for _ in range(4):
        for pt in seq:
            pt.take_steps(3)
        for pt in seq:
            pt.swap()
    par = ladders()
    random.seed(5)
    evals = advance_ladders(par, 12, swap_interval=3, batch_posterior=batch)
    assert evals >= 2 * 3 * 12 * 3  # at least one evaluation per chain, step and parameter
    for a, b in zip(seq, par):
        for ca, cb in zip(a.chains, b.chains):
            assert np.array_equal(ca.get_sample(burn=0), cb.get_sample(burn=0))
            assert np.array_equal(np.array(ca.probs), np.array(cb.probs))
        assert np.array_equal(a.successful_swaps, b.successful_swaps)


@pytest.mark.parametrize("temps,n,si", [([1.0, 2.0, 4.0, 8.0], 14, 4), ([1.0, 3.0, 9.0], 9, 3), ([1.0, 1.5, 2.5, 4.0, 7.0, 12.0], 10, 5)])
def test_pairwise_swaps_equal_the_ladder_run_alone(golden, problem, temps, n, si):
    """Ladders with generators of their own (the sharded / config-5 form): every PAIR swaps as soon as its two chains have
    finished the interval and goes on, while the rest of the ladder is still in it (advance_ladders, round 4) - sample
    paths, probabilities and swap counts must be those of `ParallelTempering.advance` on each ladder alone, trailing
    partial interval and a chain that sits a swap out (odd ladder) included."""
    g = golden("pt")
    gp, start, widths = problem

    def batch(thetas):
        return np.array([gp.marginal_likelihood(t) for t in thetas])

    def ladders():
        out = []
        for lad in range(3):
            chains = [make_chain(gp.marginal_likelihood, start, widths, g["hp_bounds"], t, 9000 + 100 * lad + 10 * k)
                      for k, t in enumerate(temps)]
            pt = ParallelTempering(chains, batch_posterior=None)
            pt.rng = default_rng(170 + lad)
            pt.pair_choice = random.Random(300 + lad).choice
            out.append(pt)
        return out

    alone = ladders()
    for pt in alone:
        pt.advance(n, swap_interval=si)
    together = ladders()
    advance_ladders(together, n, swap_interval=si, batch_posterior=batch)
    for a, b in zip(alone, together):
        for ca, cb in zip(a.chains, b.chains):
            assert np.array_equal(ca.get_sample(burn=0), cb.get_sample(burn=0))
            assert np.array_equal(np.array(ca.probs), np.array(cb.probs))
        assert np.array_equal(a.successful_swaps, b.successful_swaps)
        assert np.array_equal(a.attempted_swaps, b.attempted_swaps)
        assert a.rng.random() == b.rng.random()  # the ladders' generators have been consumed identically


def test_lockstep_on_analytic_posterior():
    def post(t):
        return float(-0.5 * np.sum((t - 1.0) ** 2 / np.array([0.5, 2.0]) ** 2))

    def batch(th):
        return np.array([post(t) for t in th])

    chains = [GibbsChain(post, np.array([0.0, 0.0]), widths=[1.0, 1.0], temperature=T) for T in (1.0, 2.0, 4.0)]
    for k, c in enumerate(chains):
        c.rng = default_rng(k)
        for i, p in enumerate(c.params):
            p.rng = default_rng(10 * k + i)
    advance_lockstep(chains, 400, batch)
    s = chains[0].get_sample(burn=100)
    assert abs(s[:, 0].mean() - 1.0) < 0.3 and abs(s[:, 1].mean() - 1.0) < 0.8
    assert chains[0].chain_length == 401


def test_posterior_validation():
    with pytest.raises(ValueError):
        GibbsChain(posterior=3.0, start=np.zeros(2))
    with pytest.raises(ValueError):
        GibbsChain(posterior=lambda t: 1, start=np.zeros(2))  # int, not float (base.py:277)
    with pytest.raises(ValueError):
        GibbsChain(posterior=lambda t: float("nan"), start=np.zeros(2))
