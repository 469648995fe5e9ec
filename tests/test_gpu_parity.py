"""
GPU parity tests (run with `-m gpu` on the MI355X box): the HIP path, called through the
C-ABI (ctypes), against (a) the committed golden vectors produced by the imported reference and
(b) the CPU oracle on the same seeded inputs.

Tolerance (BASELINE.json north_star): 1e-10 relative error for GpRegressor outputs in fp64,
measured as max |a - b| / max |b| per output array.  Inputs carry y_err = 0.1 (cond(K) ~ 1e4-1e6);
with y_err = None the reference's only regulariser is the 1e-12 jitter (cond ~ 1e12) and no
implementation — including the reference under a different BLAS — can agree to 1e-10.
"""
import warnings

import numpy as np
import pytest

import workloads as wl

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def check(a, b, tol=RTOL, what=""):
    r = rel(a, b)
    _record(what, r, tol)
    assert r <= tol, f"{what}: relative error {r:.3e} > {tol:.1e}"
    return r


def _record(what, r, tol):
    """Every comparison is logged; conftest.py prints the achieved errors at the end of the run, so that a
    regression of a couple of digits is visible long before it reaches a tolerance."""
    import inspect

    test = next((f.function for f in inspect.stack() if f.function.startswith("test_")), "?")
    ACHIEVED.append((test, what, r, tol))


ACHIEVED = []
PT_SAMPLE_TOL = 1e-12  # sample paths of the teacher-forced chains: not equality - the proposal-width adaptation feeds the
# device likelihood's last digits back into the samples (gibbs.py:132-148); measured 0 .. 4.8e-15 (profiles/r05_parity_errors.txt)


def check_each(a, b, tol=RTOL, what="", floor=1e-6, etol=1e-7):
    """Normwise `check` plus an element-wise one for vectors whose small entries matter (alpha, gradient
    components): the normwise measure leaves entries far below the maximum unconstrained, so every element larger
    than `floor` x the largest is also held to `etol` relative to ITSELF.  An entry at 1e-6 of the maximum
    inherits a relative error of normwise-error / 1e-6 from the cancellation that made it small (measured: 2.4e-8
    at N = 6500, 1e-11 .. 1e-12 on the small cases), hence etol = 1e-7 and not 1e-10."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    check(a, b, tol, what)
    big = np.abs(b) > floor * np.abs(b).max()
    r = float((np.abs(a - b)[big] / np.abs(b)[big]).max()) if big.any() else 0.0
    _record(what + " (element-wise)", r, etol)
    assert r <= etol, f"{what}: element-wise relative error {r:.3e} > {etol:.1e}"
    return r


@pytest.fixture(scope="module")
def gp_mod():
    from inference_amd import gp

    return gp


def kernel_cls(gp_mod, kid):
    return gp_mod.SquaredExponential if kid == wl.SE else gp_mod.RationalQuadratic


# ---------------------------------------------------------------------------------------
# golden vectors of the reference's own test data (N = 32, d = 2)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,kid", [("se_", wl.SE), ("rq_", wl.RQ)])
def test_t32_fit_and_predict_vs_reference(golden, gp_mod, name, kid):
    g = golden("t32")
    th = g[name + "thetas"]
    gp = gp_mod.GpRegressor(g["x"], g["y"], y_err=g["y_err"], hyperpars=th[0], kernel=kernel_cls(gp_mod, kid))
    assert [str(s) for s in g[name + "labels"]] == gp.hyperpar_labels
    assert str(g[name + "str"]) == str(gp)
    check(np.array(gp.hp_bounds), g[name + "hp_bounds"], 1e-12, "hp_bounds")
    check(gp.K_xx, g[name + "K_xx"], 1e-13, "K_xx")
    # the reference's own 32-point test set: measured 1e-14 (L) .. 1e-13 (alpha, mu, sigma); held to 2e-12 so that
    # a 1e-11 error in the factor is caught here (test_suite_detects_a_1e11_fault)
    check(gp.L, g[name + "L"], 2e-12, what="L")
    assert np.all(np.triu(gp.L, 1) == 0.0)
    check_each(gp.alpha, g[name + "alpha"], 2e-12, what="alpha")
    mu, sig = gp(g[name + "pts"])
    check(mu, g[name + "mu"], 2e-12, what="mu")
    check(sig, g[name + "sig"], 2e-12, what="sig")
    pm, pc = gp.build_posterior(g[name + "pts"][:16])
    check(pm, g[name + "post_mu"], what="posterior mean")
    check(pc, g[name + "post_cov"], what="posterior covariance")
    check(gp.build_posterior(g[name + "pts"][:16], mean_only=True), g[name + "post_mu"])
    lml = [gp.marginal_likelihood(t) for t in th]
    check(lml, g[name + "lml"], what="lml")
    check(gp.marginal_likelihood_batch(th), g[name + "lml"], what="lml batch")
    # kernel plugin surface
    check(gp.cov.build_covariance(th[1][1:]), g[name + "cov_K"], 1e-13, "build_covariance")
    check(gp.cov(g[name + "pts"], g["x"], th[1][1:]), g[name + "cov_cross"], 1e-13, "cov.__call__")


def test_t32_white_noise_composite(golden, gp_mod):
    g = golden("t32")
    th = g["sewn_theta"]
    gp = gp_mod.GpRegressor(
        g["x"], g["y"], y_err=g["y_err"], hyperpars=th, kernel=gp_mod.SquaredExponential() + gp_mod.WhiteNoise()
    )
    assert [str(s) for s in g["sewn_labels"]] == gp.hyperpar_labels
    check(np.array(gp.hp_bounds), g["sewn_hp_bounds"], 1e-12)
    check(gp.marginal_likelihood(th), g["sewn_lml"])
    check_each(gp.alpha, g["sewn_alpha"], what="alpha")
    mu, sig = gp(g["se_pts"])
    check(mu, g["sewn_mu"])
    check(sig, g["sewn_sig"])


def test_t32_one_dimensional_input(golden, gp_mod):
    g = golden("t32")
    gp = gp_mod.GpRegressor(g["d1_x"], g["d1_y"], y_err=g["d1_err"], hyperpars=g["d1_theta"])
    check(np.array(gp.hp_bounds), g["d1_hp_bounds"], 1e-12)
    mu, sig = gp(g["d1_pts"])
    check(mu, g["d1_mu"])
    check(sig, g["d1_sig"])


# ---------------------------------------------------------------------------------------
# BASELINE configs against golden vectors (reference) and the oracle
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["cfg1", "rq256", "cfg4", "cfg2"])
def test_config_golden(golden, gp_mod, case):
    g = golden(case)
    cfg, kid, n, d = [int(v) for v in g["meta"]]
    x, y, e = wl.synthetic_dataset(cfg, n, d)
    th = g["thetas"]
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=th[0], kernel=kernel_cls(gp_mod, kid))
    check(np.array(gp.hp_bounds), g["hp_bounds"], 1e-12, "hp_bounds")
    check(gp.marginal_likelihood_batch(th), g["lml"], what="lml")
    check(np.linalg.norm(gp.alpha), g["alpha_norm"], what="|alpha|")
    check_each(gp.alpha[g["alpha_idx"]], g["alpha_sub"], what="alpha")
    check(gp._logdet, g["logdet"], what="logdet")
    mu, sig = gp(g["pts"])
    check(mu, g["mu"], what="mu")
    check(sig, g["sig"], what="sig")
    pm, pc = gp.build_posterior(g["pts"][:16])
    check(pm, g["post_mu"], what="posterior mean")
    check(pc, g["post_cov"], what="posterior cov")
    if n <= 4096:
        check(np.diagonal(gp.L)[g["alpha_idx"]], g["diagL_sub"], what="diag L")


@pytest.mark.parametrize("n,d,kid", [(2, 1, wl.SE), (5, 3, wl.SE), (127, 2, wl.RQ), (129, 4, wl.SE), (640, 16, wl.RQ), (1000, 7, wl.SE)])
def test_ragged_sizes_vs_oracle(gp_mod, n, d, kid):
    """Sizes that are not multiples of the 128 tile (padding paths), tiny and odd shapes."""
    from oracle import gp_oracle as orc

    x, y, e = wl.synthetic_dataset(50 + n, n, d)
    th = wl.timing_theta(kid, y, d) 
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=th, kernel=kernel_cls(gp_mod, kid))
    ref = orc.OracleGp(x, y, e, kernel=kid, hyperpars=th)
    check(gp.K_xx, ref.K_xx, 1e-13, "K")
    check(gp.L, ref.L, what="L")
    check_each(gp.alpha, ref.alpha, what="alpha")
    check(gp.marginal_likelihood(th), ref.marginal_likelihood(th), what="lml")
    pts = wl.query_points(n, 37, d)
    mu, sig = gp(pts)
    rmu, rsig = ref(pts)
    check(mu, rmu, what="mu")
    check(sig, rsig, what="sig")


def test_cholesky_failure_sentinel(golden, gp_mod):
    """Indefinite y_cov: LinAlgError -> warn + -1e50 in marginal_likelihood (regression.py:540-542)."""
    g = golden("fail")
    gp = gp_mod.GpRegressor(g["x"], g["y"], y_cov=g["y_cov"], hyperpars=g["theta_ok"])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        v = gp.marginal_likelihood(g["theta_bad"])
    assert v == -1e50 and any("Cholesky" in str(x.message) for x in w)
    check(gp.marginal_likelihood(g["theta_ok"]), g["lml_ok"])
    with pytest.raises(np.linalg.LinAlgError):
        gp.set_hyperparameters(g["theta_bad"])


def test_input_consistency_checking(gp_mod):
    """tests/gp/test_GpRegressor.py:154-160 of the reference."""
    with pytest.raises(ValueError):
        gp_mod.GpRegressor(x=np.zeros(3), y=np.zeros(2))
    with pytest.raises(ValueError):
        gp_mod.GpRegressor(x=np.zeros([4, 3]), y=np.zeros(3))
    with pytest.raises(ValueError):
        gp_mod.GpRegressor(x=np.zeros([3, 1]), y=np.zeros([3, 2]))


def test_runs_are_bit_reproducible(gp_mod):
    """Deterministic reductions: the same call twice gives identical bits."""
    x, y, e = wl.synthetic_dataset(9, 700, 5)
    th = wl.timing_theta(wl.SE, y, 5)
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=th)
    a1, l1 = gp.alpha.copy(), gp.marginal_likelihood(th)
    gp.set_hyperparameters(th)
    assert np.array_equal(a1, gp.alpha) and l1 == gp.marginal_likelihood(th)


# ---------------------------------------------------------------------------------------
# LML gradient (regression.py:544-567) and leave-one-out (regression.py:451-487)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,kid", [("se_", wl.SE), ("rq_", wl.RQ)])
def test_t32_lml_gradient_and_loo_vs_reference(golden, gp_mod, name, kid):
    g = golden("t32")
    th = g[name + "thetas"]
    gp = gp_mod.GpRegressor(g["x"], g["y"], y_err=g["y_err"], hyperpars=th[0], kernel=kernel_cls(gp_mod, kid))
    res = [gp.marginal_likelihood_gradient(t) for t in th]
    check([r[0] for r in res], g[name + "lml_g_val"], what="lml (grad path)")
    for r, ref in zip(res, g[name + "lml_g_grad"]):
        check_each(r[1], ref, what="lml gradient")
    lm, ls = gp.loo_predictions()
    check(lm, g[name + "loo_mu"], what="loo mu")
    check(ls, g[name + "loo_sig"], what="loo sigma")
    check([gp.loo_likelihood(t) for t in th], g[name + "loo"], what="loo likelihood")
    # the fitted state must be untouched by the scratch-lane evaluations
    check(gp.alpha, g[name + "alpha"], what="alpha after scratch work")
    mu, _ = gp(g[name + "pts"])
    check(mu, g[name + "mu"])


def test_t32_white_noise_gradient(golden, gp_mod):
    g = golden("t32")
    th = g["sewn_theta"]
    gp = gp_mod.GpRegressor(
        g["x"], g["y"], y_err=g["y_err"], hyperpars=th, kernel=gp_mod.SquaredExponential() + gp_mod.WhiteNoise()
    )
    v, grad = gp.marginal_likelihood_gradient(th)
    check(v, g["sewn_lml_g_val"])
    check_each(grad, g["sewn_lml_g_grad"], what="gradient incl. WhiteNoise parameter")


@pytest.mark.parametrize("case", ["cfg1", "rq256", "cfg4"])
def test_config_lml_gradient_golden(golden, gp_mod, case):
    g = golden(case)
    cfg, kid, n, d = [int(v) for v in g["meta"]]
    x, y, e = wl.synthetic_dataset(cfg, n, d)
    th = g["thetas"]
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=th[0], kernel=kernel_cls(gp_mod, kid))
    k = min(len(th), 4)
    res = [gp.marginal_likelihood_gradient(t) for t in th[:k]]
    check([r[0] for r in res], g["lml_g_val"][:k], what="lml")
    # gradient components: relative to the largest component of each gradient vector
    for r, ref in zip(res, g["lml_g_grad"][:k]):
        check_each(r[1], ref, what="lml gradient")


def test_lml_gradient_matches_finite_differences(gp_mod):
    """The reference's own property test (tests/gp/test_GpRegressor.py:61-76) on the device path."""
    rng = np.random.default_rng(1)
    points = rng.uniform(low=0.0, high=2.0, size=(32, 2))
    values = np.sin(points[:, 0]) * np.cos(points[:, 1]) + rng.normal(scale=0.1, size=32)
    gp = gp_mod.GpRegressor(points, values, y_err=np.full(32, 0.1), hyperpars=np.array([0.0, -0.5, 0.5, 0.5]))
    trng = np.random.default_rng(123)
    for theta in trng.uniform(low=[-0.3, -1.5, 0.1, 0.1], high=[0.3, 0.5, 1.5, 1.5], size=[10, 4]):
        _, grad = gp.marginal_likelihood_gradient(theta)
        fd = np.zeros(4)
        for i in range(4):
            dx = theta[i] * 1e-5
            t1, t2 = theta.copy(), theta.copy()
            t1[i] -= dx
            t2[i] += dx
            fd[i] = 0.5 * (gp.marginal_likelihood(t2) - gp.marginal_likelihood(t1)) / dx
        assert abs(fd / grad - 1.0).max() < 1e-5


def test_hyperparameter_search_bfgs_and_diffev(gp_mod):
    """tests/gp/test_GpRegressor.py:147-151: the optimisers run end to end on the device objective
    and land on a hyper-parameter vector at least as good as the oracle's fit from the same start."""
    rng = np.random.default_rng(1)
    points = rng.uniform(low=0.0, high=2.0, size=(32, 2))
    values = np.sin(points[:, 0]) * np.cos(points[:, 1]) + rng.normal(scale=0.1, size=32)
    errs = np.full(32, 0.1)
    np.random.seed(3)
    gp = gp_mod.GpRegressor(points, values, y_err=errs, optimizer="bfgs", n_starts=4)
    centre = np.array([0.5 * (a + b) for a, b in gp.hp_bounds])
    assert gp.marginal_likelihood(gp.hyperpars) >= gp.marginal_likelihood(centre) - 1e-9
    gp2 = gp_mod.GpRegressor(points, values, y_err=errs, optimizer="diffev")
    assert abs(gp2.marginal_likelihood(gp2.hyperpars) - gp.marginal_likelihood(gp.hyperpars)) < 1e-3
    mu, sig = gp(points)
    assert np.isfinite(mu).all() and np.isfinite(sig).all()


# ---------------------------------------------------------------------------------------
# spatial gradients (regression.py:351-419) and acquisition functions (acquisition.py:44-232)
# ---------------------------------------------------------------------------------------
def test_t32_spatial_gradients_vs_reference(golden, gp_mod):
    g = golden("t32")
    gp = gp_mod.GpRegressor(g["x"], g["y"], y_err=g["y_err"], hyperpars=g["se_thetas"][0])
    pts = g["se_pts"]
    gm, gc = gp.gradient(pts[:16])
    check(gm, g["se_grad_mu"], what="gradient mean")
    check(gc, g["se_grad_cov"], what="gradient covariance")
    sm, sv = gp.spatial_derivatives(pts[:16])
    check(sm, g["se_sd_mu"], what="d mu / dx")
    check(sv, g["se_sd_var"], what="d var / dx")
    # 1-D input: squeezed shapes (regression.py:385, 419)
    gp1 = gp_mod.GpRegressor(g["d1_x"], g["d1_y"], y_err=g["d1_err"], hyperpars=g["d1_theta"])
    gm, gc = gp1.gradient(g["d1_pts"])
    assert gm.shape == g["d1_grad_mu"].shape and gc.shape == g["d1_grad_cov"].shape
    check(gm, g["d1_grad_mu"])
    check(gc, g["d1_grad_cov"])
    sm, sv = gp1.spatial_derivatives(g["d1_pts"])
    check(sm, g["d1_sd_mu"])
    check(sv, g["d1_sd_var"])
    # RationalQuadratic has no gradient_terms: same NotImplementedError as the reference
    gq = gp_mod.GpRegressor(g["x"], g["y"], y_err=g["y_err"], hyperpars=g["rq_thetas"][0], kernel=gp_mod.RationalQuadratic)
    with pytest.raises(NotImplementedError):
        gq.gradient(pts[:2])
    with pytest.raises(NotImplementedError):
        gq.spatial_derivatives(pts[:2])


@pytest.mark.parametrize("nm", ["ei", "ucb", "mv"])
def test_t32_acquisition_vs_reference(golden, gp_mod, nm):
    g = golden("t32")
    gp = gp_mod.GpRegressor(g["x"], g["y"], y_err=g["y_err"], hyperpars=g["se_thetas"][0])
    acq = {"ei": gp_mod.ExpectedImprovement, "ucb": gp_mod.UpperConfidenceBound, "mv": gp_mod.MaxVariance}[nm]()
    acq.update_gp(gp)
    pts = g["se_pts"]
    check([acq(p) for p in pts[:6]], g[f"se_{nm}_call"][:6], what="__call__")
    check(acq.call_batch(pts), g[f"se_{nm}_call"], what="call_batch")
    check([acq.opt_func(p) for p in pts[:6]], g[f"se_{nm}_opt"][:6], what="opt_func")
    check(acq.opt_func_batch(pts), g[f"se_{nm}_opt"], what="opt_func_batch")
    val, grad = acq.opt_func_gradient_batch(pts)
    check(val, g[f"se_{nm}_optg_val"], what="opt_func_gradient value")
    check(grad, g[f"se_{nm}_optg_grad"], what="opt_func_gradient gradient")
    v1, g1 = acq.opt_func_gradient(pts[3])
    check(float(v1), g[f"se_{nm}_optg_val"][3])
    check(g1, g[f"se_{nm}_optg_grad"][3], what="opt_func_gradient gradient (scalar API)")
    check([acq.convergence_metric(p) for p in pts[:4]], g[f"se_{nm}_conv"], what="convergence metric")


def test_cfg4_expected_improvement_1000_candidates(golden, gp_mod):
    """BASELINE config 4: N=4096, d=4, 1000 EI candidates in one batched evaluation, both branches
    of the Z < -3 switch (acquisition.py:79, 91, 104)."""
    g = golden("cfg4")
    x, y, e = wl.synthetic_dataset(4, 4096, 4)
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=g["thetas"][0])
    ei = gp_mod.ExpectedImprovement()
    ei.update_gp(gp)
    check(ei.mu_max, g["mu_max"], 0.0)
    check(ei.call_batch(g["cand"]), g["ei_call"], what="EI")
    check(ei.opt_func_batch(g["cand"]), g["ei_opt"], what="-ln EI")
    val, grad = ei.opt_func_gradient_batch(g["cand"][:200])
    check(val, g["ei_optg_val"], what="-ln EI (gradient call)")
    check(grad, g["ei_optg_grad"], 2e-10, "grad -ln EI")
    ei.mu_max = float(g["ei_far_mu_max"])  # forces Z < -3
    # EI itself is exp(ln EI) with ln EI ~ -100 here: its relative error is |ln EI| times that of ln EI (measured 1.2e-10)
    check(ei.call_batch(g["cand"][:200]), g["ei_far_call"], 1e-9, "EI (Z < -3)")
    check(ei.opt_func_batch(g["cand"][:200]), g["ei_far_opt"], what="-ln EI (Z < -3)")
    val, grad = ei.opt_func_gradient_batch(g["cand"][:200])
    check(val, g["ei_far_optg_val"], what="-ln EI (Z < -3, gradient call)")
    check(grad, g["ei_far_optg_grad"], 2e-10, "grad -ln EI (Z < -3)")


@pytest.mark.parametrize("acq", ["ei", "ucb", "mv"])
def test_gp_optimiser_loop(gp_mod, acq):
    """tests/gp/test_GpOptimiser.py:21-68 of the reference: propose / add iterations stay in bounds."""
    acquisition = {"ei": gp_mod.ExpectedImprovement, "ucb": gp_mod.UpperConfidenceBound, "mv": gp_mod.MaxVariance}[acq]

    def objective(x):
        return np.sin(0.5 * x[0]) * 3 / (2 + 0.5 * (x[1] - 1.0) ** 2) + 0.1 * x[0]

    rng = np.random.default_rng(4)
    bounds = [(-4.0, 6.0), (-3.0, 5.0)]
    x = rng.uniform([b[0] for b in bounds], [b[1] for b in bounds], size=(8, 2))
    y = np.array([objective(k) for k in x])
    np.random.seed(11)
    opt = gp_mod.GpOptimiser(x, y, bounds=bounds, acquisition=acquisition)
    for _ in range(3):
        new_x = opt.propose_evaluation()
        assert len(new_x) == 2
        assert all(b[0] <= v <= b[1] for v, b in zip(new_x, bounds))
        opt.add_evaluation(new_x, objective(new_x))
    assert opt.y.size == 11 and len(opt.iteration_history) == 3
    # 1-D problem with errors and the differential-evolution proposal path
    x1 = np.linspace(-3, 3, 7)
    y1 = np.cos(x1) + 0.1 * x1
    opt1 = gp_mod.GpOptimiser(x1, y1, bounds=[(-4.0, 4.0)], y_err=np.full(7, 0.05), acquisition=acquisition)
    nx = opt1.propose_evaluation(optimizer="diffev")
    assert -4.0 <= float(nx) <= 4.0
    opt1.add_evaluation(nx, np.cos(nx) + 0.1 * nx, new_y_err=0.05)
    assert opt1.y.size == 8


# ---------------------------------------------------------------------------------------
# MCMC callers on the device posterior (configs 3 and 5)
# ---------------------------------------------------------------------------------------
def _pt_problem():
    rng = np.random.default_rng(31)
    x = np.sort(rng.uniform(0, 6, 48))
    y = np.sin(x) + 0.3 * np.cos(2.5 * x) + 0.2 * rng.normal(size=48)
    return x, y, np.full(48, 0.2), np.array([y.mean(), np.log(y.std()), np.log(1.0)]), np.array([0.1, 0.2, 0.2])


def test_parallel_tempering_on_device_reproduces_reference_trace(golden, gp_mod):
    """Teacher-forced ParallelTempering (parallel.py:190-281) over GibbsChains whose posterior is the
    DEVICE marginal_likelihood, advanced in lockstep (batched LML evaluations): same accept / swap
    decisions and sample paths as the reference's process-per-chain run."""
    import random
    from numpy.random import default_rng
    from inference_amd.mcmc import GibbsChain, ParallelTempering

    g = golden("pt")
    x, y, e, start, widths = _pt_problem()
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=start)
    gp.engine.set_streams(4)
    check(np.array(gp.hp_bounds), g["hp_bounds"], 1e-12)

    def make_chain(temp, seed):
        ch = GibbsChain(posterior=gp.marginal_likelihood, start=start, widths=widths, temperature=temp)
        for i, b in enumerate(gp.hp_bounds):
            ch.set_boundaries(i, b)
        ch.rng = default_rng(seed)
        for i, par in enumerate(ch.params):
            par.rng = default_rng(seed + 1 + i)
        return ch

    ch = make_chain(1.0, 100)
    ch.advance(60)
    # accept / reject decisions are bit-determined; the sample VALUES inherit the device likelihood's last digits
    # through the proposal-width adaptation (gibbs.py:132-148), so they are held to a tolerance, not to equality
    check(np.asarray(ch.get_sample(burn=0)), g["single_samples"], PT_SAMPLE_TOL, "single chain samples")
    check(np.array(ch.probs), g["single_probs"], 1e-10)

    chains = [make_chain(t, 1000 + 10 * k) for k, t in enumerate(g["temps"])]
    pt = ParallelTempering(chains)
    assert pt.batch_posterior is not None  # detected GpRegressor.marginal_likelihood -> batched lockstep
    pt.rng = default_rng(7)
    random.seed(9)
    pt.advance(40, swap_interval=5)
    for k, c in enumerate(pt.return_chains()):
        check(np.asarray(c.get_sample(burn=0)), g[f"pt_samples_{k}"], PT_SAMPLE_TOL, f"tempered chain {k} samples")
        check(np.array(c.probs), g[f"pt_probs_{k}"], 1e-10)
    assert np.array_equal(pt.successful_swaps, g["pt_successful"])
    assert pt.posterior_evaluations >= 4 * 40 * 3


def test_sweep_single_rank(gp_mod):
    """Config 3 shape at reduced size: RQ, d = 16, an 8 x 8 theta grid through the sharded sweep
    (one rank here; the 2-rank gather is covered by tests/test_sharding_cpu.py)."""
    from inference_amd import sharding
    from oracle import gp_oracle as orc

    x, y, e = wl.synthetic_dataset(3, 384, 16)
    grid = wl.theta_grid_cfg3(y, 16)
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=grid[0], kernel=gp_mod.RationalQuadratic)
    gp.engine.set_streams(4)
    vals = sharding.marginal_likelihood_sweep(gp, grid)
    ref = orc.OracleGp(x, y, e, kernel=orc.RQ)
    want = np.array([ref.marginal_likelihood(t) for t in grid])
    check(vals, want, what="64-point RQ sweep")
    # a checksum of the sweep is invariant under the order in which streams finish
    assert vals.sum() == sharding.marginal_likelihood_sweep(gp, grid).sum()


def test_t32_loo_gradient_vs_reference(golden, gp_mod):
    """loo_likelihood_gradient (regression.py:489-526) and cross_val=True hyper-parameter selection."""
    g = golden("t32")
    for name, kid in (("se_", wl.SE), ("rq_", wl.RQ)):
        th = g[name + "thetas"]
        gp = gp_mod.GpRegressor(g["x"], g["y"], y_err=g["y_err"], hyperpars=th[0], kernel=kernel_cls(gp_mod, kid))
        res = [gp.loo_likelihood_gradient(t) for t in th]
        check([r[0] for r in res], g[name + "loo_g_val"], what="LOO value")
        for r, ref in zip(res, g[name + "loo_g_grad"]):
            check_each(r[1], ref, what="LOO gradient")
    # finite-difference property of the reference's own test (tests/gp/test_GpRegressor.py:79-94)
    theta = g["se_thetas"][2]
    gp = gp_mod.GpRegressor(g["x"], g["y"], y_err=g["y_err"], hyperpars=theta)
    _, grad = gp.loo_likelihood_gradient(theta)
    for i in range(theta.size):
        dx = theta[i] * 1e-5
        t1, t2 = theta.copy(), theta.copy()
        t1[i] -= dx
        t2[i] += dx
        fd = 0.5 * (gp.loo_likelihood(t2) - gp.loo_likelihood(t1)) / dx
        assert abs(fd / grad[i] - 1.0) < 1e-5
    np.random.seed(2)
    gp_cv = gp_mod.GpRegressor(g["x"], g["y"], y_err=g["y_err"], cross_val=True, n_starts=3)
    assert gp_cv.loo_likelihood(gp_cv.hyperpars) >= gp_cv.loo_likelihood(theta) - 1e-6


def test_batched_lockstep_equals_single_evaluations(gp_mod):
    """gpmi_lml_batch's lockstep path (blockIdx.z batch, N <= 4096) against one-at-a-time evaluations:
    different kernels / tile shapes, same numbers to 1e-12; ragged batch sizes."""
    x, y, e = wl.synthetic_dataset(12, 300, 3)
    thetas = wl.theta_set(wl.RQ, y, 3, 37)
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=thetas[0], kernel=gp_mod.RationalQuadratic)
    single = np.array([gp.marginal_likelihood(t) for t in thetas])
    for b in (37, 5, 2):
        check(gp.marginal_likelihood_batch(thetas[:b]), single[:b], 1e-12, f"batch of {b}")
    # a failing member does not disturb its neighbours
    bad = thetas.copy()
    bad[3, 1] = 800.0  # amplitude overflow -> non-finite pivot
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vals = gp.marginal_likelihood_batch(bad[:6])
    assert vals[3] == -1e50
    check(np.delete(vals, 3), np.delete(single[:6], 3), 1e-12)


@pytest.mark.parametrize("kid,wn,mean", [(wl.SE, False, "const"), (wl.RQ, True, "const"), (wl.SE, False, "linear")])
def test_batched_gradient_equals_single_evaluations(gp_mod, kid, wn, mean):
    """gpmi_lml_grad_batch (lockstep: K-build, factorisation, both sweeps, L^-T, the k-skipped SYRK and the fused
    contraction all carry the batch in blockIdx.z) against one-at-a-time `marginal_likelihood_gradient` calls: value,
    every gradient component (mean, amplitude, shape, length scales, WhiteNoise), ragged batch sizes; and the multi-start
    search driven in lockstep off it finds the optimum of the serial search."""
    n, d = 700, 3
    x, y, e = wl.synthetic_dataset(21, n, d)
    cov = kernel_cls(gp_mod, kid)()
    if wn:
        cov = cov + gp_mod.WhiteNoise()
    kw = dict(kernel=cov)
    if mean == "linear":
        kw["mean"] = gp_mod.LinearMean
    base = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=None, n_starts=1, **kw) if False else None
    thetas = wl.theta_set(kid, y, d, 9)
    if wn:
        thetas = np.hstack([thetas, np.linspace(-3.0, -1.0, len(thetas))[:, None]])
    if mean == "linear":
        rng = np.random.default_rng(4)
        thetas = np.hstack([thetas[:, :1], 0.2 * rng.standard_normal((len(thetas), d)), thetas[:, 1:]])
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=thetas[0], **kw)
    single = [gp.marginal_likelihood_gradient(t) for t in thetas]
    for b in (9, 4, 1):
        f, g = gp.marginal_likelihood_gradient_batch(thetas[:b])
        check(f, [r[0] for r in single[:b]], 1e-12, f"LML, batch of {b}")
        for k in range(b):
            check_each(g[k], single[k][1], 1e-11, what=f"gradient, batch of {b}")
    if kid == wl.SE and mean == "const":
        np.random.seed(5)
        a = gp_mod.GpRegressor(x, y, y_err=e, n_starts=4, **kw)  # lockstep search
        assert a._lockstep_search()
        np.random.seed(5)
        starts_ref = gp_mod.GpRegressor.__new__(gp_mod.GpRegressor)
        b_ = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=thetas[0], **kw)
        np.random.seed(5)
        lwr, upr = [np.array([k[i] for k in b_.hp_bounds]) for i in (0, 1)]
        starts = [lwr + (upr - lwr) * np.random.random(size=len(b_.hp_bounds)) for _ in range(3)] + [0.5 * (lwr + upr)]
        serial = sorted((b_.launch_bfgs(s0) for s0 in starts), key=lambda r: r[1])[0]
        check(a.marginal_likelihood(a.hyperpars), -serial[1], 1e-9, "LML at the optimum: lockstep vs serial search")
        check(a.hyperpars, serial[0], 1e-6, "theta*: lockstep vs serial search")


@pytest.mark.parametrize("kid,wn,mean", [(wl.SE, False, "const"), (wl.RQ, True, "const"), (wl.SE, True, "linear")])
def test_batched_loo_gradient_equals_single_evaluations(gp_mod, kid, wn, mean):
    """gpmi_loo_grad_batch (lockstep: K-build, factorisation, sweeps, L^-T, diag(K^-1), the two SYRKs, the LOO vectors, the
    fused contraction all with the batch in blockIdx.z) against one-at-a-time `loo_likelihood_gradient` calls
    (regression.py:489-526): value and every gradient component, ragged batch sizes."""
    n, d = 600, 3
    x, y, e = wl.synthetic_dataset(23, n, d)
    cov = kernel_cls(gp_mod, kid)()
    if wn:
        cov = cov + gp_mod.WhiteNoise()
    kw = dict(kernel=cov)
    if mean == "linear":
        kw["mean"] = gp_mod.LinearMean
    thetas = wl.theta_set(kid, y, d, 7)
    if wn:
        thetas = np.hstack([thetas, np.linspace(-3.0, -1.0, len(thetas))[:, None]])
    if mean == "linear":
        rng = np.random.default_rng(4)
        thetas = np.hstack([thetas[:, :1], 0.2 * rng.standard_normal((len(thetas), d)), thetas[:, 1:]])
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=thetas[0], cross_val=True, **kw)
    single = [gp.loo_likelihood_gradient(t) for t in thetas]
    for b in (7, 3, 1):
        f, g = gp.loo_likelihood_gradient_batch(thetas[:b])
        check(f, [r[0] for r in single[:b]], 1e-11, f"LOO likelihood, batch of {b}")
        for k in range(b):
            check_each(g[k], single[k][1], 1e-10, what=f"LOO gradient, batch of {b}", etol=1e-6)
    if kid == wl.SE and not wn:
        # the cross-validation search in lockstep ends where the start-by-start search ends
        np.random.seed(5)
        a = gp_mod.GpRegressor(x, y, y_err=e, cross_val=True, n_starts=3, **kw)
        assert a._lockstep_search()
        starts = [l[0] for l in a.search_log]
        serial = sorted((a.launch_bfgs(s0) for s0 in starts), key=lambda r: r[1])[0]
        check(a.loo_likelihood(a.hyperpars), -serial[1], 1e-8, "LOO at the optimum: lockstep vs serial search")


def test_rccl_gather_single_rank(gp_mod):
    """gpmi_comm_* with world = 1 (the only RCCL configuration a 1-GPU box offers): unique id,
    communicator, all-gather through the library's own stream."""
    from inference_amd import sharding

    x, y, e = wl.synthetic_dataset(13, 64, 2)
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=wl.timing_theta(wl.SE, y, 2))
    eng = gp.engine
    eng.comm_init(0, 1, eng.comm_unique_id())
    out = eng.comm_allgather(np.array([1.5, -2.0, 3.25]))
    assert out.shape == (1, 3) and np.array_equal(out[0], [1.5, -2.0, 3.25])
    vals = sharding.sharded_map(lambda th: gp.marginal_likelihood_batch(th), wl.theta_set(wl.SE, y, 2, 5), engine=eng)
    assert vals.shape == (5, 1)


def test_rccl_broadcast_and_count_single_rank():
    """gpmi_comm_broadcast / gpmi_comm_count on a communicator of its own (DeviceComm: no data set needed - this is what
    runs before a rank has any), world = 1: the one RCCL configuration of a 1-GPU box.  The N > 1 Python path is covered on
    CPU (tests/test_sharding_cpu.py::test_broadcast_dataset_two_ranks_gloo)."""
    from inference_amd import sharding
    from inference_amd._engine import DeviceComm

    comm = DeviceComm()
    try:
        comm.comm_init(0, 1, comm.comm_unique_id())
        assert comm.comm_count() == 1
        v = np.arange(7.0) * 0.5 - 1.0
        assert np.array_equal(comm.comm_broadcast(v, 0), v)
        assert np.array_equal(comm.comm_allgather(v)[0], v)
        x, y, e = wl.synthetic_dataset(13, 64, 2)
        gx, gy, ge = sharding.broadcast_dataset(x, y, e, comm=comm)
        assert np.array_equal(gx, x) and np.array_equal(gy, y) and np.array_equal(ge, e)
        with pytest.raises(Exception):
            comm.comm_broadcast(v, 3)  # root outside the communicator
    finally:
        comm.comm_destroy()
        comm.close()


def test_two_ranks_on_one_device(tmp_path):
    """What the first multi-GPU run does first, on the hardware there is: two fresh processes on ONE device run
    `marginal_likelihood_sweep`, `multistart_sweep` and `tempering_run` on device engines through a FileRendezvous
    (tests/sharded_rank.py; RCCL refuses two ranks on one device, so the gather is the rendezvous-file path) and must
    reproduce the single-process results bit for bit - blocks of uneven size included."""
    import os
    import secrets
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tests", "sharded_rank.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    single = str(tmp_path / "single.npz")
    run = subprocess.run([sys.executable, tool, single], env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    key = f"test_{os.getpid()}_{secrets.token_hex(4)}"
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), WORLD_SIZE="2", GPMI_RDV_KEY=key, GPMI_RDV_DIR=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, tool, str(tmp_path / f"rank{r}.npz")], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    ref = dict(np.load(single))
    for r in range(2):
        got = dict(np.load(str(tmp_path / f"rank{r}.npz")))
        for k in ref:
            assert np.array_equal(got[k], ref[k]), (r, k)


@pytest.mark.parametrize("case,n", [("cfg2g", 8192), ("head16kg", 16384)])
def test_gradients_above_4096_vs_reference(golden, gp_mod, case, n):
    """The LML and LOO gradients at the sizes their timings are quoted for.  Above N = 4096 they take a chain of kernels of
    their own (one lane, L^-T by TRSM on the identity, the k-skipped SYRK, the fused trace pass: api_regression.hip), which
    until round 4 was only checked to N = 6500.  cfg2g = the imported reference at BASELINE config 2's size (N = 8192,
    regression.py:468-526,544-567); head16kg = the metric's own size, from the oracle's one-matrix-at-a-time gradients,
    themselves pinned to the reference in tests/test_oracle_golden.py (N = 32 ... 2048)."""
    g = golden(case)
    d = 8
    x, y, e = wl.synthetic_dataset(2, n, d)
    th = g["thetas"]
    assert np.array_equal(th, wl.theta_set(wl.SE, y, d, 2))
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=th[0])
    for t, val, ref in zip(th, g["lml_g_val"], g["lml_g_grad"]):
        v, gr = gp.marginal_likelihood_gradient(t)
        check(v, val, what=f"lml (gradient path), N = {n}")
        check_each(gr, ref, what=f"lml gradient, N = {n}")
    v, gr = gp.loo_likelihood_gradient(th[1])
    check(v, g["loo_g_val"][0], what=f"loo (gradient path), N = {n}")
    check_each(gr, g["loo_g_grad"][0], what=f"loo gradient, N = {n}")
    if "loo" in g:
        check(gp.loo_likelihood(th[1]), g["loo"][0], what=f"loo likelihood, N = {n}")
        lm, ls = gp.loo_predictions()  # at the fitted th[0]
        check(lm[g["loo_idx"]], g["loo_mu_sub"], what=f"loo means, N = {n}")
        check(ls[g["loo_idx"]], g["loo_sig_sub"], what=f"loo sigmas, N = {n}")


def test_size_32768_identities(gp_mod):
    """One size above the metric's: SE, N = 32768, d = 8 (8.6 GB per matrix; the reference has no limit but RAM,
    regression.py:241).  Proves tile indices, workspaces and the flag-ordered tail's hand-over past N = 16384 through
    identities whose right-hand sides are formed on the host from the oracle's kernel function, without downloading L:
    (i) rows of K alpha = y - mu (2048 rows: the first and last 512 and 1024 drawn at random); (ii) the LML of a second,
    independent evaluation against -1/2 r.alpha - log det of the fit; (iii) mean and variance at training inputs,
    mu*(x_i) = y_i - D_ii alpha_i and var*(x_i) = D_ii - D_ii^2 (K^-1)_ii with (K^-1)_ii from the leave-one-out path."""
    from oracle import gp_oracle as orc

    n, d = 32768, 8
    x, y, e = wl.synthetic_dataset(2, n, d)
    th = wl.timing_theta(wl.SE, y, d)
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=th)
    alpha = gp.alpha.copy()
    assert np.isfinite(alpha).all()
    r = y - th[0]
    a2 = np.exp(2.0 * th[1])
    D = e**2 + 1e-12 * a2
    rows = np.unique(np.concatenate([np.arange(512), np.arange(n - 512, n),
                                     np.random.default_rng(8).choice(n, 1024, replace=False)]))
    Ka = np.empty(rows.size)
    for lo in range(0, rows.size, 256):
        sel = rows[lo:lo + 256]
        blk = orc.kernel_cross(orc.SE, x[sel], x, th[1:])
        blk[np.arange(sel.size), sel] += D[sel]
        Ka[lo:lo + 256] = (blk * alpha[None, :]).sum(axis=1)
    check(Ka, r[rows], 1e-10, "K alpha = y - mu at N = 32768 (2048 rows)")
    lml = gp.marginal_likelihood(th)
    ident = -0.5 * float((r * alpha).sum()) - gp._logdet
    assert abs(lml - ident) <= 1e-10 * abs(lml), (lml, ident)
    _record("LML identity at N = 32768", abs(lml - ident) / abs(lml), 1e-10)
    idx = np.random.default_rng(9).choice(n, 256, replace=False)
    mu, sig = gp(x[idx])
    check(mu, y[idx] - D[idx] * alpha[idx], 1e-10, "mean at training inputs, N = 32768")
    _, loo_sig = gp.loo_predictions()
    var_expected = D[idx] - D[idx] ** 2 / loo_sig[idx] ** 2
    assert np.abs(sig**2 - var_expected).max() <= 1e-10 * a2
    gp.engine.close()


def test_bench_eight_ranks_on_one_device(tmp_path):
    """The driver's 8-GPU command on the hardware there is: `bench.py --gpus 8` with all eight ranks on ONE device
    (reduced sizes).  The ranks must agree that they share a device (physical identity, not the visible index), skip
    RCCL, gather through the rendezvous files, report both sharded configurations, and leave no rendezvous directory."""
    import glob
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    env.update(BENCH_CFG3_POINTS="8", BENCH_CFG5_LADDERS="8", BENCH_CFG5_STEPS="4", BENCH_CFG3_N="4096",
               GPMI_RDV_DIR=str(tmp_path), MASTER_PORT="29533", GPMI_DEBUG_INFO="1")
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                          "--n", "4096", "--m", "256"], env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["steps"] == 2
    par = line["config"]["parallelism"]
    assert "file-fallback" in par and "share a device" in par, par
    assert "rccl_ranks_seen" not in line["config"]  # nobody claims a communicator that was never made
    assert line["config"]["gather_schedule"] == "once_at_end"
    assert line["config"]["rank_cpu"]["cores_per_rank"] >= 1
    sh = line["sharded"]
    assert "error" not in sh and sh["config3"]["lml_evals_per_s"] > 0 and sh["config5"]["lml_evals_per_s"] > 0, (sh, [l for l in run.stderr.splitlines() if "[gpmi]" in l or "Error" in l or "line " in l][-40:])
    assert np.isfinite(line["value"]) and line["value"] > 0
    assert glob.glob(str(tmp_path / "*")) == [], "the rendezvous directory is removed at the end"


# ---------------------------------------------------------------------------------------
# BASELINE.json's full sizes: the oracle needs minutes there, so the checks are identities
# that hold at any size, with every right-hand side formed on the host from the oracle's kernels
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg,n,d,kid", [(2, 16384, 8, wl.SE), (3, 16384, 16, wl.RQ)])
def test_full_size_identities(gp_mod, cfg, n, d, kid):
    """Headline workload (SE, N = 16384, d = 8) and config 3 (RQ, N = 16384, d = 16):
    (i) K alpha = y - mu with K rebuilt row-chunk by row-chunk by the oracle's kernel functions;
    (ii) LML = -1/2 (y - mu).alpha - sum ln L_ii with the diagonal of the downloaded factor;
    (iii) 64 rows of L L^T against the oracle's K;
    (iv) the predictive mean at training inputs: mu*(x_i) = y_i - D_ii alpha_i, D = y_err^2 + 1e-12 a^2;
    (v) the predictive variance there: var*(x_i) = D_ii - D_ii^2 (K^-1)_ii, with (K^-1)_ii from the
        leave-one-out path (a different chain of kernels: L^-T by TRSM on the identity + row norms)."""
    from oracle import gp_oracle as orc

    x, y, e = wl.synthetic_dataset(cfg, n, d)
    th = wl.timing_theta(kid, y, d)
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=th, kernel=kernel_cls(gp_mod, kid))
    alpha = gp.alpha.copy()
    r = y - th[0]
    a2 = np.exp(2.0 * th[1])
    D = e**2 + 1e-12 * a2

    # (i) + (iii): rows of K from the oracle, 512 at a time (cross-covariance carries no jitter / noise)
    rows_checked = np.sort(np.random.default_rng(5).choice(n, 64, replace=False))
    Krows = np.empty((64, n))
    Ka = np.empty(n)
    for lo in range(0, n, 512):
        hi = min(lo + 512, n)
        blk = orc.kernel_cross(kid, x[lo:hi], x, th[1:])
        blk[np.arange(hi - lo), np.arange(lo, hi)] += D[lo:hi]
        Ka[lo:hi] = (blk * alpha[None, :]).sum(axis=1)
        sel = (rows_checked >= lo) & (rows_checked < hi)
        Krows[sel] = blk[rows_checked[sel] - lo]
    check(Ka, r, 1e-10, "K alpha = y - mu")

    L = gp.L
    logdet = float(np.log(np.diagonal(L)).sum())
    lml = gp.marginal_likelihood(th)
    assert abs(lml - (-0.5 * float((r * alpha).sum()) - logdet)) <= 1e-10 * abs(lml)
    LLt = np.einsum("rk,nk->rn", L[rows_checked], L)  # rows of L L^T (L is lower triangular)
    check(LLt, Krows, 1e-10, "L L^T = K")
    del L, LLt

    idx = np.random.default_rng(6).choice(n, 256, replace=False)
    mu, sig = gp(x[idx])
    check(mu, y[idx] - D[idx] * alpha[idx], 1e-10, "mean at training inputs")
    _, loo_sig = gp.loo_predictions()  # sigma_loo^2 = 1 / (K^-1)_ii   (regression.py:466)
    ik = 1.0 / loo_sig[idx] ** 2
    var_expected = D[idx] - D[idx] ** 2 * ik
    # var* is a difference of O(a^2) numbers: 1e-10 relative to a^2, the scale the kernels work at
    assert np.abs(sig**2 - var_expected).max() <= 1e-10 * a2


def test_headline_size_against_the_oracle(golden, gp_mod):
    """The configuration BASELINE.json's metric is quoted on (SE, N = 16384, d = 8) against values of the row-chunked
    oracle (tests/golden/head16k.npz; the oracle itself is pinned to the imported reference up to N = 8192, K bit for
    bit): alpha at 64 indices and its norm, diag(L), log det, LML at three hyper-parameter vectors, mean / sigma at 64
    query points, the posterior at 16 (regression.py:218-244, 188-216, 421-449, 528-542)."""
    g = golden("head16k")
    n, d = 16384, 8
    x, y, e = wl.synthetic_dataset(2, n, d)
    th = g["thetas"]
    assert np.array_equal(th[0], wl.timing_theta(wl.SE, y, d)) and np.array_equal(th, wl.theta_set(wl.SE, y, d, 3))
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=th[0])
    ii = g["alpha_idx"]
    check_each(gp.alpha[ii], g["alpha_sub"], what="alpha at 64 indices")
    check(np.linalg.norm(gp.alpha), g["alpha_norm"], what="|alpha|")
    check(gp._logdet, g["logdet"], what="log det")
    mu, sig = gp(g["pts"])
    check(mu, g["mu"], what="mean")
    check(sig, g["sig"], what="sigma")
    pm, pc = gp.build_posterior(g["pts"][:16])
    check(pm, g["post_mu"], what="posterior mean")
    check(pc, g["post_cov"], what="posterior covariance")
    check(np.diagonal(gp.L)[ii], g["diagL_sub"], what="diag(L) at 64 indices")
    check([gp.marginal_likelihood(t) for t in th], g["lml"], what="LML at 3 thetas")
    check(gp.marginal_likelihood_batch(th), g["lml"], what="LML batch")


def test_config3_share_of_the_grid_against_the_oracle(golden, gp_mod):
    """BASELINE config 3 at full size (RQ, N = 16384, d = 16): the 8 grid points of one GPU's share of the 64-point
    sweep through `marginal_likelihood_sweep` with 4 lanes, against the row-chunked oracle (tests/golden/cfg3_16k.npz);
    fit + predict at the first of them (regression.py:528-542; covariance.py:343-348)."""
    from inference_amd import sharding

    g = golden("cfg3_16k")
    n, d = 16384, 16
    x, y, e = wl.synthetic_dataset(3, n, d)
    grid = wl.theta_grid_cfg3(y, d)
    sel = g["grid_idx"]
    assert np.array_equal(grid[sel], g["thetas"])
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=grid[sel[0]], kernel=gp_mod.RationalQuadratic)
    check_each(gp.alpha[g["alpha_idx"]], g["alpha_sub"], what="alpha at 64 indices")
    check(np.linalg.norm(gp.alpha), g["alpha_norm"], what="|alpha|")
    check(gp._logdet, g["logdet"], what="log det")
    mu, sig = gp(g["pts"])
    check(mu, g["mu"], what="mean")
    check(sig, g["sig"], what="sigma")
    gp.engine.set_streams(4)
    vals = sharding.marginal_likelihood_sweep(gp, grid[sel])
    check(vals, g["lml"], what="8-point share of the RQ grid")
    assert np.array_equal(vals, sharding.marginal_likelihood_sweep(gp, grid[sel]))


# ---------------------------------------------------------------------------------------
# GpLinearInverter: device path vs the reference's golden vectors
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("prob", ["deconv", "tomo"])
@pytest.mark.parametrize("tag,kid,wn", [("se", wl.SE, False), ("rq", wl.RQ, False), ("sewn", wl.SE, True)])
def test_linear_inverter_vs_reference(golden, gp_mod, prob, tag, kid, wn):
    """LML, its gradient, posterior mean and covariance at three hyper-parameter vectors.  The posterior is
    computed in Woodbury form (Cholesky solves) while the reference uses an LU solve of I + K A^T S^-1 A
    (inversion.py:150-155): the two agree to the conditioning of that non-symmetric system, hence 2e-9
    relative for the posterior mean (measured 2e-10), 2e-10 for its covariance (measured 2e-11) and 1e-10 for the
    likelihood path (measured 1e-11 on the gradient)."""
    g = golden("linv")
    pos, A, y, y_err = wl.linv_problem(prob)
    cov = kernel_cls(gp_mod, kid)()
    if wn:
        cov = cov + gp_mod.WhiteNoise()
    gli = gp_mod.GpLinearInverter(y=y, y_err=y_err, model_matrix=A, parameter_spatial_positions=pos,
                                  prior_covariance_function=cov)
    assert list(g[f"{prob}_{tag}_labels"]) == gli.hyperpar_labels
    idx = np.arange(0, 400, 7)
    for i in range(3):
        key = f"{prob}_{tag}_{i}"
        th = g[key + "_theta"]
        check(gli.marginal_likelihood(th), g[key + "_lml"], what="lml")
        lml, grad = gli.marginal_likelihood_gradient(th)
        check(lml, g[key + "_lml2"], what="lml (gradient call)")
        check_each(grad, g[key + "_grad"], what="lml gradient")
        pm, pc = gli.calculate_posterior(th)
        check(pm, g[key + "_pmean"], 2e-9, "posterior mean")
        check(gli.calculate_posterior_mean(th), g[key + "_pmean_only"], 2e-9, "posterior mean only")
        check(pc if prob == "deconv" else pc[idx][:, idx], g[key + "_pcov"], 2e-10, "posterior covariance")


def test_linear_inverter_optimise_and_errors(gp_mod):
    """optimize_hyperparameters (host Nelder-Mead over device likelihoods) improves the evidence; the
    constructor's shape checks raise as in the reference; unsupported priors are refused."""
    pos, A, y, y_err = wl.linv_problem("deconv")
    gli = gp_mod.GpLinearInverter(y=y, y_err=y_err, model_matrix=A, parameter_spatial_positions=pos)
    start = np.array([0.1, 0.0, np.log(0.15)])
    best = gli.optimize_hyperparameters(start)
    assert gli.marginal_likelihood(best) > gli.marginal_likelihood(start)
    mean = gli.calculate_posterior_mean(best)
    assert np.abs(A @ mean - y).max() < 0.2
    with pytest.raises(ValueError):
        gp_mod.GpLinearInverter(y=y, y_err=y_err, model_matrix=A[:-1], parameter_spatial_positions=pos)
    with pytest.raises(ValueError):
        gp_mod.GpLinearInverter(y=y, y_err=y_err, model_matrix=A, parameter_spatial_positions=pos[:-1])
    with pytest.raises(ValueError):
        gli.optimize_hyperparameters(np.zeros(5))
    # a prior without a device kernel is served by the dense entry points, not refused
    both = gp_mod.GpLinearInverter(y=y, y_err=y_err, model_matrix=A, parameter_spatial_positions=pos,
                                   prior_covariance_function=gp_mod.SquaredExponential() + gp_mod.RationalQuadratic())
    assert both._dense and np.isfinite(both.marginal_likelihood(np.array([0.1, 0.0, np.log(0.15), -1.0, 0.0, np.log(0.3)])))


def test_linear_inverter_with_any_covariance_object(golden, gp_mod):
    """`prior_covariance_function` may be any CovarianceFunction (inversion.py:117-127): a user-defined Matern-3/2 written
    against the plugin ABC only (tomography problem, 300 x 400) and ChangePoint over [SE, RQ] (deconvolution), against
    the reference running the same objects (tests/golden/linvp.npz).  The host evaluates the object's own
    build_covariance / covariance_and_gradients; A K A^T + Sigma, the factorisation, the solves, J^-1 and A^T J^-1 A run on
    the device (gpmi_linv_*_dense)."""
    from inference_amd.gp.covariance import CovarianceFunction

    class Matern32(wl.Matern32Math, CovarianceFunction):
        pass

    g = golden("linvp")
    pos, A, y, y_err = wl.linv_problem("tomo")
    gli = gp_mod.GpLinearInverter(y=y, y_err=y_err, model_matrix=A, parameter_spatial_positions=pos,
                                  prior_covariance_function=Matern32())
    assert list(g["m32_labels"]) == gli.hyperpar_labels
    th = g["m32_thetas"]
    check([gli.marginal_likelihood(t) for t in th], g["m32_lml"], what="lml (plugin prior)")
    res = [gli.marginal_likelihood_gradient(t) for t in th]
    check([r[0] for r in res], g["m32_lml2"], what="lml (gradient call)")
    for r, ref in zip(res, g["m32_grad"]):
        check_each(r[1], ref, what="lml gradient (plugin prior)")
    idx = np.arange(0, 400, 7)
    pm, pc = gli.calculate_posterior(th[0])
    check(pm, g["m32_pmean"], 2e-9, "posterior mean")
    check(pc[idx][:, idx], g["m32_pcov"], 2e-10, "posterior covariance")
    check(gli.calculate_posterior_mean(th[0]), g["m32_pmean_only"], 2e-9, "posterior mean only")
    pos, A, y, y_err = wl.linv_problem("deconv")
    cp = gp_mod.ChangePoint(kernels=[gp_mod.SquaredExponential, gp_mod.RationalQuadratic])
    gli = gp_mod.GpLinearInverter(y=y, y_err=y_err, model_matrix=A, parameter_spatial_positions=pos,
                                  prior_covariance_function=cp)
    assert list(g["cp_labels"]) == gli.hyperpar_labels
    th = g["cp_theta"]
    check(gli.marginal_likelihood(th), g["cp_lml"], what="lml (ChangePoint prior)")
    lml, grad = gli.marginal_likelihood_gradient(th)
    check(lml, g["cp_lml2"])
    check_each(grad, g["cp_grad"], what="lml gradient (ChangePoint prior)")
    pm, pc = gli.calculate_posterior(th)
    check(pm, g["cp_pmean"], 2e-9, "posterior mean")
    check(pc, g["cp_pcov"], 2e-10, "posterior covariance")


# ---------------------------------------------------------------------------------------
# HeteroscedasticNoise: per-point noise hyper-parameters
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,with_err", [("err", True), ("noerr", False)])
def test_heteroscedastic_noise_vs_reference(golden, gp_mod, tag, with_err):
    """SE + HeteroscedasticNoise, N = 96 (99 hyper-parameters): labels, bounds, LML, the full gradient (the 96
    noise components from one device vector), fit (K, alpha) and predict against the reference."""
    g = golden("het")
    n, d = 96, 1
    x, y, e = wl.synthetic_dataset(77, n, d)
    th = g[f"{tag}_thetas"]
    gp = gp_mod.GpRegressor(x, y, y_err=e if with_err else None, hyperpars=th[1],
                            kernel=gp_mod.SquaredExponential() + gp_mod.HeteroscedasticNoise())
    assert list(g[f"{tag}_labels"]) == gp.hyperpar_labels
    check(np.array(gp.hp_bounds, dtype=float), g[f"{tag}_bounds"], 1e-12, "bounds")
    check_each(gp.alpha, g[f"{tag}_alpha"], what="alpha")
    check(gp.marginal_likelihood_batch(th), g[f"{tag}_lml"], what="lml")
    res = [gp.marginal_likelihood_gradient(t) for t in th]
    check([r[0] for r in res], g[f"{tag}_lml2"], what="lml (gradient call)")
    for r, ref in zip(res, g[f"{tag}_grad"]):
        check_each(r[1], ref, what="gradient")
    # the same through the lockstep batch (gpmi_lml_grad_batch_noise: every evaluation carries noise variances of its own):
    # against the reference's values, and the multi-start search may now run its starts together
    fb, gb = gp.marginal_likelihood_gradient_batch(th)
    check(fb, g[f"{tag}_lml2"], what="lml (batched gradient call)")
    for row, ref in zip(gb, g[f"{tag}_grad"]):
        check_each(row, ref, what="gradient (batched)")
    assert gp._lockstep_search()
    check(gp.K_xx, g[f"{tag}_K_xx"], 1e-13, "K_xx after evaluations at other thetas")
    mu, sig = gp(wl.query_points(77, 40, d))
    check(mu, g[f"{tag}_mu"], what="mu")
    check(sig, g[f"{tag}_sig"], what="sig")
    check([gp.loo_likelihood(t) for t in th], g[f"{tag}_loo"], what="loo likelihood")


# ---------------------------------------------------------------------------------------
# ChangePoint: mixture of stationary kernels with per-point weights
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,subs,wn", [("sese", (wl.SE, wl.SE), False), ("serq", (wl.SE, wl.RQ), False),
                                          ("sesewn", (wl.SE, wl.SE), True)])
def test_change_point_vs_reference(golden, gp_mod, tag, subs, wn):
    """Labels, bounds, fit (K_xx, alpha), predict, LML and the full LML gradient (sub-kernel parameters, change-point
    location / width, WhiteNoise) and the leave-one-out predictions against the reference."""
    g = golden("cp")
    x, y, e, pts = g["x"], g["y"], g["y_err"], g["pts"]
    th = g[f"{tag}_thetas"]
    cov = gp_mod.ChangePoint(kernels=[kernel_cls(gp_mod, k) for k in subs])
    if wn:
        cov = cov + gp_mod.WhiteNoise()
    gp = gp_mod.GpRegressor(x, y, y_err=e, kernel=cov, hyperpars=th[1])
    assert list(g[f"{tag}_labels"]) == gp.hyperpar_labels
    check(np.array(gp.hp_bounds, dtype=float), g[f"{tag}_bounds"], 1e-12, "bounds")
    check_each(gp.alpha, g[f"{tag}_alpha"], what="alpha")
    check(gp.K_xx, g[f"{tag}_K_xx"], 1e-13, "K_xx")
    mu, sig = gp(pts)
    check(mu, g[f"{tag}_mu"], what="mu")
    check(sig, g[f"{tag}_sig"], what="sig")
    check([gp.marginal_likelihood(t) for t in th], g[f"{tag}_lml"], what="lml")
    res = [gp.marginal_likelihood_gradient(t) for t in th]
    check([r[0] for r in res], g[f"{tag}_lml2"], what="lml (gradient call)")
    for r, ref in zip(res, g[f"{tag}_grad"]):
        check_each(r[1], ref, what="gradient")
    # the likelihood evaluations above used other hyper-parameters: the fitted state must be restored lazily
    mu2, sig2 = gp(pts)
    check(mu2, g[f"{tag}_mu"], what="mu after other evaluations")
    loo_mu, loo_sig = gp.loo_predictions()
    check(loo_mu, g[f"{tag}_loo_mu"], what="loo mean")
    check(loo_sig, g[f"{tag}_loo_sig"], what="loo sigma")
    pm, pc = gp.build_posterior(pts[:20])
    check(pm, g[f"{tag}_post_mu"], what="posterior mean")
    check(pc, g[f"{tag}_post_cov"], what="posterior covariance")
    check([gp.loo_likelihood(t) for t in th], g[f"{tag}_loo"], what="loo likelihood")


@pytest.mark.parametrize("tag,subs,wn", [("sese", (wl.SE, wl.SE), False), ("serq", (wl.SE, wl.RQ), False),
                                          ("sesewn", (wl.SE, wl.SE), True)])
def test_change_point_batched_gradient(golden, gp_mod, tag, subs, wn):
    """gpmi_lml_grad_batch_mix (round 4): the mixture's LML and full gradient for several hyper-parameter vectors in one
    lockstep call - against the reference's values (the `cp` fixture) and against one-at-a-time evaluations, for ragged
    batch sizes; the lockstep search is on for such a model and the fitted state survives the batch."""
    g = golden("cp")
    x, y, e, pts = g["x"], g["y"], g["y_err"], g["pts"]
    th = g[f"{tag}_thetas"]
    cov = gp_mod.ChangePoint(kernels=[kernel_cls(gp_mod, k) for k in subs])
    if wn:
        cov = cov + gp_mod.WhiteNoise()
    gp = gp_mod.GpRegressor(x, y, y_err=e, kernel=cov, hyperpars=th[1])
    assert gp._lockstep_search()
    rng = np.random.default_rng(8)
    more = np.vstack([th, th[0] + 0.05 * rng.standard_normal((5, th.shape[1]))])
    single = [gp.marginal_likelihood_gradient(t) for t in more]
    for b in (len(more), len(th), 1):
        f, gr = gp.marginal_likelihood_gradient_batch(more[:b])
        check(f, [r[0] for r in single[:b]], 1e-12, f"mixture LML, batch of {b}")
        for k in range(b):
            check_each(gr[k], single[k][1], 1e-11, what=f"mixture gradient, batch of {b}")
    f, gr = gp.marginal_likelihood_gradient_batch(th)
    check(f, g[f"{tag}_lml2"], what="mixture LML (batched) vs reference")
    for r, ref in zip(gr, g[f"{tag}_grad"]):
        check_each(r, ref, what="mixture gradient (batched) vs reference")
    mu, sig = gp(pts)  # the batch used the shared weight buffers: the fit is restored lazily
    check(mu, g[f"{tag}_mu"], what="mu after the batched evaluations")


@pytest.mark.parametrize("model", ["se", "change_point"])
def test_lockstep_search_is_independent_of_the_grouping(gp_mod, model):
    """(round 6) A start of the lockstep multi-start search (regression.py:585-605) ends where it ends whatever other starts
    share its rounds: the search evaluates every round - also one with a single survivor - with the lockstep kernels
    (GPMI_OPT_LOCKSTEP_ALWAYS), whose values do not depend on the batch (gemm: `ring_order_only`), so the iterates of
    six starts advanced together are, bit for bit, those of every start advanced alone.  The single-evaluation kernels
    (`launch_bfgs`, one start after another) give the same LML and a gradient within 1e-11 (a different order of summation):
    those iterates fork in the last digits on a flat optimum, which is why the two paths' optima agree to ~1e-11 only
    (profiles/r06_search.json)."""
    from inference_amd.gp._lockstep import lockstep_lbfgsb

    rng = np.random.default_rng(11)
    n = 700
    x = np.sort(rng.uniform(0, 1, n)).reshape(-1, 1)
    y = np.where(x[:, 0] < 0.5, np.sin(4 * x[:, 0]), np.sin(40 * x[:, 0])) + 0.05 * rng.normal(size=n)
    e = np.full(n, 0.05)
    if model == "se":
        th0 = np.array([0.0, 0.0, np.log(0.1)])
        gp = gp_mod.GpRegressor(x, y, y_err=e, kernel=gp_mod.SquaredExponential, hyperpars=th0)
    else:
        th0 = np.array([0.0, 0.0, np.log(0.3), 0.0, np.log(0.05), 0.5, 0.05])
        gp = gp_mod.GpRegressor(x, y, y_err=e, kernel=gp_mod.ChangePoint(kernels=[gp_mod.SquaredExponential] * 2), hyperpars=th0)
    assert gp._lockstep_search()
    lwr, upr = (np.array([b[i] for b in gp.hp_bounds], dtype=float) for i in (0, 1))
    starts = lwr + (upr - lwr) * rng.uniform(0.2, 0.8, size=(4, len(lwr)))

    def neg(X):
        f, g = gp.marginal_likelihood_gradient_batch(X)
        return -f, -g

    gp.batch_independent_values(True)
    together = lockstep_lbfgsb(neg, starts, gp.hp_bounds, maxiter=25)
    for k, x0 in enumerate(starts):
        alone = lockstep_lbfgsb(neg, x0[None, :], gp.hp_bounds, maxiter=25)[0]
        assert np.array_equal(alone[0], together[k][0]) and alone[1] == together[k][1], (k, alone[1], together[k][1])
        assert alone[2]["funcalls"] == together[k][2]["funcalls"] and alone[2]["nit"] == together[k][2]["nit"]
    # a value does not depend on its batch: the same theta alone and as the last row of a batch of five, bit for bit
    f1, g1 = gp.marginal_likelihood_gradient_batch(starts[:1])
    f5, g5 = gp.marginal_likelihood_gradient_batch(np.vstack([starts[1:], starts[:1], starts[1:2]]))
    assert f1[0] == f5[3] and np.array_equal(g1[0], g5[3])
    gp.batch_independent_values(False)
    # the single-evaluation kernels: same LML to the last bits' neighbourhood, gradient within 1e-11
    fs, gs = gp.marginal_likelihood_gradient(starts[0])
    check(fs, f1[0], 1e-13, "single-evaluation LML vs lockstep batch of one")
    check_each(gs, g1[0], 1e-11, what="single-evaluation gradient vs lockstep batch of one")
    # the constructor's search leaves the option as it found it
    np.random.seed(3)
    gp2 = gp_mod.GpRegressor(x[:200], y[:200], y_err=e[:200], kernel=gp_mod.SquaredExponential)
    assert gp2._lockstep_search() and not getattr(gp2, "_batch_independent", False)


def test_change_point_search_and_limits(gp_mod):
    """Hyper-parameter search through the mixture path (L-BFGS-B with the analytic gradient); three regions
    work for fit / predict / LML / gradient; sub-kernels without device code take the dense path."""
    rng = np.random.default_rng(11)
    x = np.sort(rng.uniform(0, 1, 120)).reshape(-1, 1)
    y = np.where(x[:, 0] < 0.5, np.sin(4 * x[:, 0]), np.sin(40 * x[:, 0])) + 0.05 * rng.normal(size=120)
    e = np.full(120, 0.05)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gp = gp_mod.GpRegressor(x, y, y_err=e, kernel=gp_mod.ChangePoint(kernels=[gp_mod.SquaredExponential] * 2))
    start = np.array([b[0] + 0.5 * (b[1] - b[0]) for b in gp.hp_bounds])
    assert gp.marginal_likelihood(gp.hyperpars) >= gp.marginal_likelihood(start)
    mu, sig = gp(x)
    assert np.abs(mu - y).mean() < 0.2
    cp3 = gp_mod.ChangePoint(kernels=[gp_mod.SquaredExponential] * 3)
    th3 = np.array([0.0, 0.0, np.log(0.3), 0.0, np.log(0.05), 0.0, np.log(0.3), 0.35, 0.03, 0.7, 0.03])
    gp3 = gp_mod.GpRegressor(x, y, y_err=e, kernel=cp3, hyperpars=th3)
    assert np.isfinite(gp3.marginal_likelihood(th3)) and np.isfinite(gp3(x[:5])[0]).all()
    # three regions: the sub-kernel components of the gradient are exact (finite differences), the window components
    # follow the reference's expression (tests/golden/cpx.npz pins them)
    _, g3 = gp3.marginal_likelihood_gradient(th3)
    for i in (1, 2, 4, 6):
        h = 1e-6
        tp, tm = th3.copy(), th3.copy()
        tp[i] += h
        tm[i] -= h
        fd = (gp3.marginal_likelihood(tp) - gp3.marginal_likelihood(tm)) / (2 * h)
        assert abs(fd - g3[i]) <= 1e-4 * max(abs(fd), 1.0)  # finite-difference noise at LML ~ -1e3 is ~2e-5
    # a combination without a fused device path (a WhiteNoise region) goes through the plugin methods + dense device path
    cpw = gp_mod.ChangePoint(kernels=[gp_mod.SquaredExponential, gp_mod.WhiteNoise])
    thw = np.array([0.0, 0.0, np.log(0.3), np.log(0.1), 0.5, 0.05])
    gpw = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=thw, kernel=cpw)
    assert gpw._generic and np.isfinite(gpw.marginal_likelihood(thw)) and np.isfinite(gpw(x[:5])[0]).all()


def test_ragged_large_size_vs_oracle(gp_mod):
    """N = 6500 (padded to 6528 = 51 tile rows): large enough for the look-ahead regime, the split trailing
    updates and the 512-wide inverse blocks with a partial last block, not a multiple of anything."""
    from oracle import gp_oracle as orc

    n, d = 6500, 5
    x, y, e = wl.synthetic_dataset(65, n, d)
    th = wl.timing_theta(wl.RQ, y, d)
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=th, kernel=gp_mod.RationalQuadratic)
    ref = orc.OracleGp(x, y, e, kernel=orc.RQ, hyperpars=th)
    check_each(gp.alpha, ref.alpha, what="alpha")
    check(gp.marginal_likelihood(th), ref.marginal_likelihood(th), what="lml")
    pts = wl.query_points(65, 300, d)
    mu, sig = gp(pts)
    rmu, rsig = ref(pts)
    check(mu, rmu, what="mu")
    check(sig, rsig, what="sig")
    lml, grad = gp.marginal_likelihood_gradient(th)
    rl, rg = ref.marginal_likelihood_gradient(th)
    check(lml, rl, what="lml (gradient call)")
    check_each(grad, rg, what="gradient")


def test_random_problems_vs_oracle():
    """A seeded random sweep (tools/fuzz_parity.py: sizes 2 .. 2600 across every tile-count boundary, d = 1 .. 11, SE / RQ,
    with and without WhiteNoise, noise scaled x 0.5 .. 3, hyper-parameters perturbed by 0.2 in every log, 1 .. 69 query
    points): fit, prediction, posterior, LML and LOO with their gradients and - SquaredExponential - the spatial
    gradients against the oracle, every quantity to 1e-10 (measured over 42 such problems: worst 2e-12, the
    posterior covariance of a d = 1 problem)."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import fuzz_parity

    res = fuzz_parity.sweep(seed=7, cases=12, nmax=2600, verbose=False)
    assert len(res) == 12
    worst = {}
    for desc, errs in res:
        for q, r in errs.items():
            assert r <= RTOL, f"{desc}: {q} relative error {r:.3e}"
            worst[q] = max(worst.get(q, 0.0), r)
    for q, r in worst.items():
        _record(f"worst {q} over 12 random problems", r, RTOL)


def test_many_query_points_vs_oracle(gp_mod):
    """Predict with more query points than training-matrix tile rows make convenient: M = 2500 (20 row tiles, ragged)
    at N = 3000 - the many-right-hand-side solve then runs its products with the inverse blocks on 64 x 64 ring tiles
    with the k-skip (few query points take the 32-row tiles), and its updates span several rounds."""
    from oracle import gp_oracle as orc

    n, d, m = 3000, 3, 2500
    x, y, e = wl.synthetic_dataset(66, n, d)
    th = wl.timing_theta(wl.SE, y, d)
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=th)
    ref = orc.OracleGp(x, y, e, kernel=orc.SE, hyperpars=th)
    pts = wl.query_points(66, m, d)
    mu, sig = gp(pts)
    rmu, rsig = ref(pts)
    check(mu, rmu, what="mu")
    check(sig, rsig, what="sig")
    few = gp(pts[:40])
    check(few[0], rmu[:40], what="mu (40 points)")
    check(few[1], rsig[:40], what="sig (40 points)")


def test_objects_pickle_without_device_state(gp_mod):
    """the reference pickles the GP object into worker processes (regression.py:600-601,
    parallel.py:130-136); here the device handle is dropped on pickling and re-created lazily."""
    import pickle

    x, y, e = wl.synthetic_dataset(9, 200, 2)
    th = wl.timing_theta(wl.SE, y, 2)
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=th)
    pts = wl.query_points(9, 10, 2)
    mu, sig = gp(pts)
    clone = pickle.loads(pickle.dumps(gp))
    assert clone._engine is None
    clone.set_hyperparameters(th)
    mu2, sig2 = clone(pts)
    assert np.array_equal(mu, mu2) and np.array_equal(sig, sig2)
    assert clone.marginal_likelihood(th) == gp.marginal_likelihood(th)
    pos, A, yy, ee = wl.linv_problem("deconv")
    gli = gp_mod.GpLinearInverter(y=yy, y_err=ee, model_matrix=A, parameter_spatial_positions=pos)
    t3 = np.array([0.1, 0.0, np.log(0.15)])
    v = gli.marginal_likelihood(t3)
    assert pickle.loads(pickle.dumps(gli)).marginal_likelihood(t3) == v


# ---------------------------------------------------------------------------------------
# the callers that draw random numbers, pinned against the reference under numpy.random.seed
# (tests/golden/search.npz): multi-start L-BFGS-B, differential evolution, acquisition starts, proposals
# ---------------------------------------------------------------------------------------
def _t32_data():
    rng = np.random.default_rng(1)
    points = rng.uniform(low=0.0, high=2.0, size=(32, 2))
    values = np.sin(points[:, 0]) * np.cos(points[:, 1]) + rng.normal(scale=0.1, size=32)
    return points, values, np.full(32, 0.1)


def _bo_problem():
    def objective(x):
        return np.sin(0.5 * x[0]) * 3 / (2 + 0.5 * (x[1] - 1.0) ** 2) + 0.1 * x[0]

    rng = np.random.default_rng(4)
    bounds = [(-4.0, 6.0), (-3.0, 5.0)]
    x = rng.uniform([b[0] for b in bounds], [b[1] for b in bounds], size=(8, 2))
    x[5] = [7.0, 1.0]
    y = np.array([objective(k) for k in x])
    return x, y, bounds


@pytest.mark.parametrize("tag,kw", [("lml", {}), ("loo", {"cross_val": True}), ("rq", {"kernel": "rq"})])
def test_multistart_bfgs_reproduces_reference_search(golden, gp_mod, tag, kw):
    """regression.py:585-605 with numpy.random.seed(3): the start positions are drawn in the reference's order
    (to the 1e-15 of the bounds), every start ends at the reference's optimum, and the selected theta* scores the same."""
    g = golden("search")
    x, y, e = _t32_data()
    kw = dict(kw)
    if kw.get("kernel") == "rq":
        kw["kernel"] = gp_mod.RationalQuadratic
    np.random.seed(3)
    gp = gp_mod.GpRegressor(x, y, y_err=e, optimizer="bfgs", n_starts=4, **kw)
    # (start, optimum, objective) per run; the runs advance in lockstep: off gpmi_lml_grad_batch (round 3) and, for the
    # cross-validation objective, off gpmi_loo_grad_batch (round 4)
    log = gp.search_log
    assert gp._lockstep_search()
    check(np.array(gp.hp_bounds, dtype=float), g[f"ms_{tag}_bounds"], 1e-12, "bounds")
    # same random numbers in the same order; the bounds they scale come from the O(N log N) identity (1e-15 apart)
    check(np.array([l[0] for l in log]), g[f"ms_{tag}_starts"], 1e-13, "start positions")
    # every L-BFGS-B run stops within its own convergence test (factr 1e7 -> ~2e-9 relative) of the reference's run
    fv = np.array([l[2] for l in log])
    err = np.abs(fv - g[f"ms_{tag}_fvals"]) / np.abs(g[f"ms_{tag}_fvals"])
    print(f"[{tag}] per-start objective agreement {err.max():.2e}; theta* distance "
          f"{np.abs(np.array(gp.hyperpars) - g[f'ms_{tag}_theta']).max():.2e}")
    assert err.max() < 1e-9
    best = float(gp.model_selector(gp.hyperpars))
    assert abs(best - float(g[f"ms_{tag}_best"])) <= 1e-8 * abs(float(g[f"ms_{tag}_best"]))
    assert np.abs(np.array(gp.hyperpars) - g[f"ms_{tag}_theta"]).max() < 1e-8  # the L-BFGS-B paths coincide (measured 2e-11)


def test_default_starts_and_differential_evolution_reproduce_reference(golden, gp_mod):
    g = golden("search")
    x, y, e = _t32_data()
    np.random.seed(8)
    gp = gp_mod.GpRegressor(x, y, y_err=e)  # int(2 sqrt(P)) + 1 starts
    v = gp.marginal_likelihood(gp.hyperpars)
    assert abs(v - float(g["ms_default_best"])) <= 1e-8 * abs(float(g["ms_default_best"]))
    np.random.seed(5)
    gpd = gp_mod.GpRegressor(x, y, y_err=e, optimizer="diffev")
    vd = gpd.marginal_likelihood(gpd.hyperpars)
    # SciPy's differential_evolution under the same global seed: the population path is identical as long as no
    # comparison flips on a 1e-13 difference; its polish step ends at the same optimum
    print(f"differential evolution: LML {vd:.12g} (reference {float(g['de_best']):.12g})")
    assert abs(vd - float(g["de_best"])) <= 1e-7 * abs(float(g["de_best"]))


def test_batched_differential_evolution_reaches_the_serial_optimum(gp_mod):
    """(round 6, extension) `GpRegressor(optimizer="diffev", diffev_batched=True)`: SciPy's differential evolution with a
    vectorised objective and deferred updating - a whole generation per lockstep device call (gpmi_lml_batch) - against the
    default form (regression.py:569-573: one evaluation per call, the reference's trajectory) on BASELINE config 1
    (SE, N = 512, d = 2), both seeded: another population walk, the same optimum (polished by L-BFGS-B in both)."""
    import workloads as wl

    x, y, e = wl.synthetic_dataset(1, 512, 2)
    np.random.seed(5)
    serial = gp_mod.GpRegressor(x, y, y_err=e, optimizer="diffev")
    np.random.seed(5)
    batched = gp_mod.GpRegressor(x, y, y_err=e, optimizer="diffev", diffev_batched=True)
    vs, vb = serial.marginal_likelihood(serial.hyperpars), batched.marginal_likelihood(batched.hyperpars)
    print(f"differential evolution: serial LML {vs:.12g}, batched LML {vb:.12g}")
    check(vb, vs, 1e-6, "batched differential evolution optimum vs serial")
    assert np.abs(np.array(batched.hyperpars) - np.array(serial.hyperpars)).max() < 1e-3
    # the batched objective itself: the values of single evaluations, and -1e50 where the factorisation fails
    th = np.array([np.array(serial.hyperpars) + 0.1 * k for k in range(5)])
    check(batched.model_selector_batch(th), [serial.marginal_likelihood(t) for t in th], 1e-12, "model_selector_batch")
    # the cross-validation objective (gpmi_loo_grad_batch values)
    np.random.seed(5)
    cv = gp_mod.GpRegressor(x[:200], y[:200], y_err=e[:200], optimizer="diffev", diffev_batched=True, cross_val=True)
    np.random.seed(5)
    cvs = gp_mod.GpRegressor(x[:200], y[:200], y_err=e[:200], optimizer="diffev", cross_val=True)
    # (two different population walks that stop on SciPy's tol = 0.01 spread criterion and are polished with finite
    # differences on a flat objective: measured 1.6e-4 apart; the values themselves are checked to 1e-12 right below)
    check(cv.loo_likelihood(cv.hyperpars), cvs.loo_likelihood(cvs.hyperpars), 2e-3, "batched DE, leave-one-out objective")
    th = np.array([np.array(cvs.hyperpars) + 0.05 * k for k in range(4)])
    check(cv.model_selector_batch(th), [cvs.loo_likelihood(t) for t in th], 1e-12, "model_selector_batch (leave-one-out)")


@pytest.mark.parametrize("nm", ["ei", "ucb", "mv"])
def test_acquisition_starts_and_proposals_reproduce_reference(golden, gp_mod, nm):
    """acquisition.py:13-37 and optimisation.py:202-249 under numpy.random.seed: the starting positions (20 probes
    per training point, ranked by one batched device evaluation; a uniform draw for the point outside the bounds) are
    the reference's, and the proposal reaches the reference's acquisition value."""
    g = golden("search")
    bx, by, bounds = _bo_problem()
    acq = {"ei": gp_mod.ExpectedImprovement, "ucb": gp_mod.UpperConfidenceBound, "mv": gp_mod.MaxVariance}[nm]
    opt = gp_mod.GpOptimiser(bx, by, bounds=bounds, hyperpars=g["bo_theta"], acquisition=acq)
    np.random.seed(21)
    starts = np.array(opt.acquisition.starting_positions(bounds))
    check(starts, g[f"bo_{nm}_starts"], 1e-13, "starting positions")
    np.random.seed(22)
    prop = np.array(opt.propose_evaluation())
    val = float(opt.acquisition.opt_func(prop))
    ref_val = float(g[f"bo_{nm}_value"])
    print(f"[{nm}] proposal distance {np.abs(prop - g[f'bo_{nm}_proposal']).max():.2e}, value {val:.12g} vs {ref_val:.12g}")
    assert abs(val - ref_val) <= 1e-8 * max(abs(ref_val), 1.0)
    assert np.abs(prop - g[f"bo_{nm}_proposal"]).max() < 1e-9  # measured <= 4e-15
    np.random.seed(23)
    prop_de = np.array(opt.propose_evaluation(optimizer="diffev"))
    val_de = float(opt.acquisition.opt_func(prop_de))
    assert abs(val_de - float(g[f"bo_{nm}_de_value"])) <= 1e-6 * max(abs(float(g[f"bo_{nm}_de_value"])), 1.0)


def test_gp_optimiser_fit_reproduces_reference(golden, gp_mod):
    g = golden("search")
    bx, by, bounds = _bo_problem()
    np.random.seed(31)
    opt = gp_mod.GpOptimiser(bx, by, bounds=bounds)
    v = opt.gp.marginal_likelihood(opt.gp.hyperpars)
    assert abs(v - float(g["bo_fit_lml"])) <= 1e-8 * abs(float(g["bo_fit_lml"]))


# ---------------------------------------------------------------------------------------
# LinearMean / QuadraticMean (mean.py:54-126): the non-constant mean path (a mean vector per evaluation)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("mtag", ["lin", "quad"])
@pytest.mark.parametrize("ktag,kid", [("se", wl.SE), ("rq", wl.RQ)])
def test_linear_and_quadratic_mean_vs_reference(golden, gp_mod, mtag, ktag, kid):
    g = golden("means")
    key = f"{mtag}_{ktag}"
    mean = {"lin": gp_mod.LinearMean, "quad": gp_mod.QuadraticMean}[mtag]
    th = g[key + "_thetas"]
    gp = gp_mod.GpRegressor(g["x"], g["y"], y_err=g["y_err"], hyperpars=th[0], kernel=kernel_cls(gp_mod, kid), mean=mean)
    assert list(g[key + "_labels"]) == gp.hyperpar_labels
    check(np.array(gp.hp_bounds, dtype=float), g[key + "_bounds"], 1e-12, "bounds")
    check(gp.mu, g[key + "_mu_train"], 1e-13, "prior mean at the training points")
    check_each(gp.alpha, g[key + "_alpha"], what="alpha")
    mu, sig = gp(g["pts"])
    check(mu, g[key + "_mu"], what="mu")
    check(sig, g[key + "_sig"], what="sig")
    pm, pc = gp.build_posterior(g["pts"][:10])
    check(pm, g[key + "_post_mu"], what="posterior mean")
    check(pc, g[key + "_post_cov"], what="posterior covariance")
    check([gp.marginal_likelihood(t) for t in th], g[key + "_lml"], what="lml")
    check(gp.marginal_likelihood_batch(th), g[key + "_lml"], what="lml batch (one mean vector per evaluation)")
    res = [gp.marginal_likelihood_gradient(t) for t in th]
    check([r[0] for r in res], g[key + "_lml2"], what="lml (gradient call)")
    for r, ref in zip(res, g[key + "_grad"]):
        check_each(r[1], ref, what="lml gradient incl. the mean-parameter components")
    check([gp.loo_likelihood(t) for t in th], g[key + "_loo"], what="loo likelihood")
    for t, ref in zip(th, g[key + "_loo_grad"]):
        check_each(gp.loo_likelihood_gradient(t)[1], ref, what="loo gradient")
    if kid == wl.SE:
        sm, sv = gp.spatial_derivatives(g["pts"][:10])
        check(sm, g[key + "_sd_mu"], what="d mu / dx")
        check(sv, g[key + "_sd_var"], what="d var / dx")


# ---------------------------------------------------------------------------------------
# dense plugin methods vs the fused device contraction
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,kid", [("se_", wl.SE), ("rq_", wl.RQ)])
def test_dense_covariance_and_gradients_vs_reference_and_device_contraction(golden, gp_mod, name, kid):
    """`covariance_and_gradients` (covariance.py:268-276, 350-365) element by element against the reference's dense
    matrices, and the LML gradient formed from those dense matrices on the host (regression.py:559-565) against the
    fused on-device contraction that never stores dK."""
    g = golden("t32")
    th = g[name + "thetas"]
    gp = gp_mod.GpRegressor(g["x"], g["y"], y_err=g["y_err"], hyperpars=th[0], kernel=kernel_cls(gp_mod, kid))
    K, dK = gp.cov.covariance_and_gradients(th[1][1:])
    check(K, g[name + "cov_K"], 1e-13, "K")
    check(np.array(dK), g[name + "cov_dK"], 1e-13, "dK / dtheta")
    lml, grad = gp.marginal_likelihood_gradient(th[1])
    Kxx = K + gp.sig
    iK = np.linalg.inv(Kxx)
    alpha = iK @ (g["y"] - th[1][0])
    Q = np.outer(alpha, alpha) - iK
    host = np.array([0.5 * (Q * d.T).sum() for d in dK])
    check_each(grad[1:], host, what="device contraction vs dense dK")


def test_suite_detects_a_1e11_fault():
    """The tolerances are tight enough to matter: with the factored diagonal blocks scaled by (1 + 1e-11) inside the
    library (GPMI_FAULT_DIAG_EPS, a test hook) the golden-vector comparisons of this file must fail."""
    import os
    import subprocess
    import sys

    here = os.path.abspath(__file__)
    env = dict(os.environ, GPMI_FAULT_DIAG_EPS="1e-11")
    run = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider",
                          "-k", "test_t32_fit_and_predict_vs_reference or test_config_golden"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode != 0, "a 1e-11 perturbation of potrf_diag's output went unnoticed:\n" + run.stdout[-2000:]
    assert "relative error" in run.stdout


def test_config5_ladders_in_lockstep_at_stated_shape(gp_mod):
    """BASELINE config 5 at its per-GPU shape: 8 ladders x 8 temperatures (64 chains) over the LML of an N = 2048,
    d = 4 GP, advanced by `advance_ladders` (one batched device evaluation per proposal round of all 64 chains,
    ragged retries included).  The trajectories are bit-identical to the same ladders advanced one at a time (8
    chains per round), i.e. a chain does not see which other chains share its batches - which is what makes the
    ladder-sharded multi-GPU run (sharding.tempering_run) independent of the number of ranks.
    Reference: gibbs.py:627-656, parallel.py:190-281."""
    import time

    from inference_amd.mcmc import advance_ladders

    n, d = 2048, 4
    x, y, e = wl.synthetic_dataset(5, n, d)
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=wl.timing_theta(wl.SE, y, d))
    gp.batch_independent_values(True)
    steps, interval = 4, 2
    together = [wl.cfg5_ladder(gp, k) for k in range(8)]
    assert all(lad.batch_posterior is not None for lad in together)
    t0 = time.perf_counter()
    evals = advance_ladders(together, steps, swap_interval=interval)
    dt = time.perf_counter() - t0
    print(f"config 5 per-GPU shape: {evals} LML evaluations in {dt:.2f} s = {evals / dt:.0f} evals/s, "
          f"{64 * steps / dt:.1f} chain steps/s")
    assert evals >= 64 * steps * gp.n_hyperpars
    alone = [wl.cfg5_ladder(gp, k) for k in range(8)]
    for lad in alone:
        lad.advance(steps, swap_interval=interval)
    for a, b in zip(together, alone):
        for ca, cb in zip(a.chains, b.chains):
            assert np.array_equal(ca.get_sample(burn=0), cb.get_sample(burn=0))
            assert np.array_equal(np.array(ca.probs), np.array(cb.probs))
        assert np.array_equal(a.successful_swaps, b.successful_swaps)
        assert np.array_equal(a.attempted_swaps, b.attempted_swaps)
    # the chains moved and stayed inside the prior box
    lo, hi = np.array(gp.hp_bounds).T
    for lad in together:
        for ch in lad.chains:
            s = ch.get_sample(burn=0)
            assert (s >= lo).all() and (s <= hi).all() and np.ptp(s, axis=0).min() > 0
    # spot check of the log-probabilities against the oracle (tempered: prob = LML / T)
    from oracle import gp_oracle as orc

    ref = orc.OracleGp(x, y, e, kernel=orc.SE)
    ch = together[3].chains[2]
    check(ch.probs[-1] / ch.inv_temp, ref.marginal_likelihood(ch.get_last()), what="LML of a chain's last position")


def test_async_batches_match_the_synchronous_call(gp_mod, monkeypatch):
    """gpmi_lml_batch_submit / gpmi_lml_batch_wait (two slots side by side) against gpmi_lml_batch: bit-identical
    values, in the order submitted; a slot cannot be submitted twice, the synchronous calls are refused while a slot
    is pending; and the tempering driver gives the same trajectories with and without the pipelined rounds."""
    from inference_amd._lib import GpmiError
    from inference_amd.mcmc import advance_ladders

    n, d = 1024, 3
    x, y, e = wl.synthetic_dataset(55, n, d)
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=wl.timing_theta(wl.SE, y, d))
    gp.batch_independent_values(True)
    assert gp.async_batches()
    rng = np.random.default_rng(3)
    base = wl.timing_theta(wl.SE, y, d)
    th = base + 0.3 * rng.normal(size=(40, len(base)))
    ref = gp.marginal_likelihood_batch(th)
    gp.marginal_likelihood_batch_submit(th[:25], 0)
    gp.marginal_likelihood_batch_submit(th[25:], 1)
    with pytest.raises(GpmiError):
        gp.marginal_likelihood_batch_submit(th[:3], 0)  # the slot is taken
    with pytest.raises(GpmiError):
        gp.marginal_likelihood_batch(th[:4])  # the workspace is in use
    b = gp.marginal_likelihood_batch_wait(1)
    a = gp.marginal_likelihood_batch_wait(0)
    assert np.array_equal(np.concatenate([a, b]), ref)
    with pytest.raises(GpmiError):
        gp.engine.h.call("gpmi_lml_batch_wait", 0, None, None)  # nothing pending
    assert np.array_equal(gp.marginal_likelihood_batch(th[:4]), ref[:4])  # and the handle is usable again
    # a mean function with per-point values (the `mus` path: T x n doubles through the pinned staging)
    gl = gp_mod.GpRegressor(x[:512], y[:512], y_err=e[:512], mean=gp_mod.LinearMean)
    tl = gl.hyperpars + 0.2 * rng.normal(size=(9, gl.n_hyperpars))
    want = gl.marginal_likelihood_batch(tl)
    gl.marginal_likelihood_batch_submit(tl[:4], 1)
    gl.marginal_likelihood_batch_submit(tl[4:], 0)
    assert np.array_equal(np.concatenate([gl.marginal_likelihood_batch_wait(1), gl.marginal_likelihood_batch_wait(0)]), want)
    # the driver: pipelined rounds (two groups of ladders through the two slots) against one batch per round
    def run(flag):
        monkeypatch.setenv("GPMI_PT_ASYNC", flag)
        lads = [wl.cfg5_ladder(gp, k, n_temps=4) for k in range(6)]
        ev = advance_ladders(lads, 4, swap_interval=2)
        return ev, lads
    ev1, l1 = run("1")
    ev0, l0 = run("0")
    assert ev1 == ev0
    for u, v in zip(l1, l0):
        for cu, cv in zip(u.chains, v.chains):
            assert np.array_equal(cu.get_sample(burn=0), cv.get_sample(burn=0))
            assert np.array_equal(np.array(cu.probs), np.array(cv.probs))
        assert np.array_equal(u.successful_swaps, v.successful_swaps)


# ---------------------------------------------------------------------------------------
# user-defined covariance functions through the plugin ABC: host builds the dense
# matrices with the plugin's own methods, the device factorises / solves (gpmi_*_dense)
# ---------------------------------------------------------------------------------------
def test_plugin_kernel_through_the_abc_vs_reference(golden, gp_mod):
    """A Matern-3/2 kernel written against `CovarianceFunction` only (workloads.Matern32Math): every public method of
    GpRegressor, the seeded hyper-parameter search and an EI proposal (analytic gradient via gradient_terms) against
    the reference running the same class (tests/golden/plugin.npz)."""
    from inference_amd.gp.covariance import CovarianceFunction

    class Matern32(wl.Matern32Math, CovarianceFunction):
        pass

    g = golden("plugin")
    x, y, e, pts, thetas = wl.plugin_problem()
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=thetas[0], kernel=Matern32)
    assert gp._generic and list(g["labels"]) == gp.hyperpar_labels
    check(np.array(gp.hp_bounds, dtype=float), g["bounds"], 1e-12, "bounds")
    check(gp.K_xx, g["K_xx"], 1e-14, "K_xx (host plugin)")
    check(gp.L, g["L"], what="L")
    check_each(gp.alpha, g["alpha"], what="alpha")
    mu, sig = gp(pts)
    check(mu, g["mu"], what="mu")
    check(sig, g["sig"], what="sig")
    pm, pc = gp.build_posterior(pts[:12])
    check(pm, g["post_mu"], what="posterior mean")
    check(pc, g["post_cov"], what="posterior covariance")
    check(gp.build_posterior(pts[:12], mean_only=True), g["post_mu"], what="posterior mean only")
    gm, gc = gp.gradient(pts[:12])
    check(gm, g["grad_mu"], what="gradient mean")
    check(gc, g["grad_cov"], what="gradient covariance")
    sm, sv = gp.spatial_derivatives(pts[:12])
    check(sm, g["sd_mu"], what="d mu / dx")
    check(sv, g["sd_var"], what="d var / dx")
    lm, ls = gp.loo_predictions()
    check(lm, g["loo_mu"], what="loo mean")
    check(ls, g["loo_sig"], what="loo sigma")
    check([gp.marginal_likelihood(t) for t in thetas], g["lml"], what="lml")
    check(gp.marginal_likelihood_batch(thetas), g["lml"], what="lml batch")
    for t, v, gr in zip(thetas, g["lml2"], g["grad"]):
        a, b = gp.marginal_likelihood_gradient(t)
        check(a, v, what="lml (gradient call)")
        check_each(b, gr, what="lml gradient")
    check([gp.loo_likelihood(t) for t in thetas], g["loo"], what="loo likelihood")
    for t, v, gr in zip(thetas, g["loo2"], g["loo_grad"]):
        a, b = gp.loo_likelihood_gradient(t)
        check(a, v, what="loo (gradient call)")
        check_each(b, gr, what="loo gradient")
    np.random.seed(17)
    gps = gp_mod.GpRegressor(x, y, y_err=e, kernel=Matern32, n_starts=3)
    v = gps.marginal_likelihood(gps.hyperpars)
    assert abs(v - float(g["search_lml"])) <= 1e-8 * abs(float(g["search_lml"]))
    bx, by, bounds = _bo_problem()
    opt = gp_mod.GpOptimiser(bx, by, bounds=bounds, hyperpars=g["bo_theta"], kernel=Matern32)
    np.random.seed(41)
    prop = np.array(opt.propose_evaluation())
    val = float(opt.acquisition.opt_func(prop))
    assert abs(val - float(g["bo_value"])) <= 1e-8 * max(abs(float(g["bo_value"])), 1.0)
    assert np.abs(prop - g["bo_proposal"]).max() < 1e-6
    # a Cholesky failure on the dense path keeps the reference's conventions (regression.py:540-542, 241)
    bad = thetas[0].copy()
    bad[1] = 400.0
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert gp.marginal_likelihood(bad) == -1e50 and any("Cholesky" in str(k.message) for k in w)


def test_four_lanes_at_full_size(gp_mod):
    """Four concurrent evaluation lanes at N = 16384 (the multi-stream path of gpmi_lml_batch that config 3 uses on
    every GPU): the single-launch triangular sweeps of different lanes are serialised through the context's gate
    (solve.hip), the values agree with one-at-a-time evaluations, twice in a row (no state left behind)."""
    n, d = 16384, 8
    x, y, e = wl.synthetic_dataset(2, n, d)
    thetas = wl.theta_set(wl.SE, y, d, 4, seed=3)
    gp = gp_mod.GpRegressor(x, y, y_err=e, hyperpars=thetas[0])
    single = np.array([gp.marginal_likelihood(t) for t in thetas])
    gp.engine.set_streams(4)
    for _ in range(2):
        check(gp.marginal_likelihood_batch(thetas), single, 1e-12, "four lanes vs one at a time")


def test_schedule_variants_agree(tmp_path):
    """The factorisation's stream choreography (look-ahead on the CU-masked pair, slices of the trailing update on the
    panel stream, covariance build split over two streams, 32-row tiles, the split point of a launch) decides where,
    when and by which tile shape a tile is computed, never from what: alpha, log-determinant and predictions of a fit at
    N = 16384 agree to rounding (1e-11: the LDS-DMA kernel and the register-staged kernels group the k of an MFMA
    differently, so the last bits move with the split points) under every setting, and the second of two consecutive
    fits in one process is bit-identical to the first.  An ordering bug between the streams (a tile updated before
    its operands are final, a slice racing the next update, a build overtaken by the first panel) is an O(1) error."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "fit_digest.py")
    variants = [
        {},
        {"GPMI_SLICE_PCT": "0"},
        {"GPMI_SLICE_PCT": "300"},
        {"GPMI_LOOKAHEAD_MIN": "0"},
        {"GPMI_LOOKAHEAD_MIN": "24", "GPMI_KBUILD_NO_SPLIT": "1"},
        {"GPMI_M32_MAX": "0", "GPMI_SPLIT_PCT": "0", "GPMI_BIG_MIN": "64"},
        {"GPMI_FLOW": "0"},  # the last 52 tile rows in stream order instead of as flag-ordered tile tasks
        {"GPMI_CHAIN_TILES": "2"},  # the tail's chain: potrf_diag + one fused launch per column (round 4) instead of ONE launch
        {"GPMI_CHAIN_TILES": "1"},  # ... two 16 x 16-tile launches behind potrf_diag
        {"GPMI_CHAIN_TILES": "0"},  # ... and the generic tile kernels
        {"GPMI_LOOKAHEAD_MIN": "84", "GPMI_FLOW_NEAR_WGS": "64", "GPMI_FLOW_NEAR": "0", "GPMI_FLOW_NEAR_D": "5"},
        # (round 6) the task lists of rounds 3-5: one list per workgroup, whole 128 x 128 chunks in q order
        {"GPMI_FLOW_SPLIT": "0", "GPMI_FLOW_QUARTER": "99999"},
        {"GPMI_FLOW_URGENT": "0", "GPMI_FLOW_QUARTER": "2"},  # chunks of the outer panels 0, 1 whole, the later ones in quarters
        {"GPMI_GEMM_MIXED": "0"},  # (round 6) the 64 x 64 remainder of a trailing update in a launch of its own
        {"GPMI_EARLY_FILL": "0"},  # (round 6) residual and the sweeps' sentinel fills behind the factorisation, not beside it
        {"GPMI_CHAIN_BN32": "0"},  # (round 6) the predict's chain products on 32 x 64 instead of 32 x 32 tiles
    ]
    base = None
    for k, extra in enumerate(variants):
        out = str(tmp_path / f"v{k}.npz")
        run = subprocess.run([sys.executable, tool, out], env=dict(os.environ, **extra), capture_output=True, text=True,
                             timeout=300)
        assert run.returncode == 0, (extra, run.stderr[-2000:])
        r = dict(np.load(out))
        for q in ("alpha", "logdet", "mu", "sig"):
            assert np.array_equal(r[q + "0"], r[q + "1"]), (extra, q, "second fit differs from the first")
        if base is None:
            base = r
            continue
        for q in ("alpha", "logdet", "mu", "sig"):
            check(r[q + "0"], base[q + "0"], 1e-11, f"{q} under {extra}")
            if "GPMI_FLOW" in extra or "GPMI_FLOW_NEAR" in extra or "GPMI_CHAIN_TILES" in extra:  # same sums: same bits
                assert np.array_equal(r[q + "0"], base[q + "0"]), (extra, q)


def test_backward_row_solve_over_wide_blocks_matches_the_narrow_steps(tmp_path, gp_mod):
    """Spatial derivatives (regression.py:388-418) need K^-1 k per query point: a forward and a BACKWARD solve with the
    factor over many right-hand sides.  Since round 6 the backward one runs over the 512-wide outer blocks (one product
    with the untransposed inverse block - k-major triangular B, contraction from the tile column's diagonal on - and one
    K = 512 update per block) instead of over 128-wide steps (GPMI_BACKWARD_OB=0, a child process: the switch is read
    once).  Ragged N with a partial last outer block, a whole number of blocks, one block only; also against the oracle."""
    import os
    import subprocess
    import sys

    from oracle import gp_oracle as orc

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys\n"
        f"sys.path[:0] = [{root!r}, {os.path.join(root, 'inference-tools_amd')!r}]\n"
        "import numpy as np, workloads as wl\n"
        "from inference_amd.gp import GpRegressor\n"
        "out = {}\n"
        "for n, d, m in ((2500, 3, 77), (4096, 4, 300), (400, 2, 9)):\n"
        "    x, y, e = wl.synthetic_dataset(4, n, d)\n"
        "    gp = GpRegressor(x, y, y_err=e, hyperpars=wl.timing_theta(wl.SE, y, d))\n"
        "    dm, dv = gp.spatial_derivatives(wl.query_points(4, m, d))\n"
        "    out[f'dm{n}'] = dm; out[f'dv{n}'] = dv\n"
        "np.savez(sys.argv[1], **out)\n")
    res = {}
    for tag, extra in (("wide", {}), ("narrow", {"GPMI_BACKWARD_OB": "0"})):
        out = str(tmp_path / f"{tag}.npz")
        run = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, **extra), capture_output=True, text=True,
                             timeout=300)
        assert run.returncode == 0, (extra, run.stderr[-2000:])
        res[tag] = dict(np.load(out))
    for k in res["wide"]:
        # (two orders of summation of the same products; the variance derivative is a difference of terms: measured 1.1e-13)
        check(res["wide"][k], res["narrow"][k], 1e-12, f"{k}: 512-wide blocks vs 128-wide steps")
    # and the wide form against the oracle at the ragged size
    n, d, m = 2500, 3, 77
    x, y, e = wl.synthetic_dataset(4, n, d)
    th = wl.timing_theta(wl.SE, y, d)
    ref = orc.OracleGp(x, y, e, kernel=wl.SE, hyperpars=th)
    dm, dv = ref.spatial_derivatives(wl.query_points(4, m, d))
    check(res["wide"]["dm2500"], dm, what="d mu / dx vs oracle")
    check(res["wide"]["dv2500"], dv, what="d var / dx vs oracle")


@pytest.mark.parametrize("n", [5200, 6500, 8192])
def test_flow_tail_is_bit_identical_to_stream_order(tmp_path, n):
    """The flag-ordered tile-task factorisation of the chain-bound part (csrc/potrf_flow.hip: persistent task kernel
    beside the bare panel chain; the whole matrix at these sizes) against the stream-ordered schedule (GPMI_FLOW=0):
    same tile bodies, same order of summation for every element, hence the SAME BITS in alpha, log det, mean and sigma -
    also with another deal of the tasks and with every matrix byte loaded at agent scope (GPMI_FLOW_PROTO=2).  A tile
    used before its inputs were final, or a stale cache line, is an O(1) difference here."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "fit_digest.py")
    res = []
    # (round 6: the default lists hold every chunk as four quarter tasks, the urgent ones in lists of their own; the fourth
    # variant is the one-list schedule of rounds 3-5 with whole 128 x 128 chunks - the ring bodies sum alike)
    for k, extra in enumerate(({"GPMI_FLOW": "0"}, {}, {"GPMI_FLOW_NEAR_WGS": "16", "GPMI_FLOW_PROTO": "2", "GPMI_FLOW_WGS": "1"},
                               {"GPMI_FLOW_SPLIT": "0", "GPMI_FLOW_QUARTER": "99999"},
                               {"GPMI_FLOW_URGENT": "99", "GPMI_FLOW_QUARTER": "1"})):
        out = str(tmp_path / f"f{k}.npz")
        run = subprocess.run([sys.executable, tool, out, str(n)], env=dict(os.environ, **extra), capture_output=True,
                             text=True, timeout=300)
        assert run.returncode == 0, (extra, run.stderr[-2000:])
        res.append(dict(np.load(out)))
    for r in res[1:]:
        for q in res[0]:
            assert np.array_equal(r[q], res[0][q]), (n, q)


def test_flow_lost_flag_is_an_error_not_a_hang(tmp_path):
    """A task that never sets its flag (test hook GPMI_FLOW_FAULT) must end in GPMI_ERR_INTERNAL within seconds - every
    poll of the task kernel and of the chain launches is bounded - and leave the GPU usable: the Python layer then
    repeats the call once with the stream-ordered schedule (GPMI_OPT_NO_FLOW), warns, and delivers the same bits."""
    import os
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "fit_digest.py")
    t0 = time.time()
    run = subprocess.run([sys.executable, tool, str(tmp_path / "x.npz"), "8192"], env=dict(os.environ, GPMI_FLOW_FAULT="7000"),
                         capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "timed out" in run.stderr and "GPMI_OPT_NO_FLOW" in run.stderr, run.stderr[-2000:]
    assert time.time() - t0 < 90
    run = subprocess.run([sys.executable, tool, str(tmp_path / "y.npz"), "8192"], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "timed out" not in run.stderr, run.stderr[-2000:]
    a, b = dict(np.load(tmp_path / "x.npz")), dict(np.load(tmp_path / "y.npz"))
    assert all(np.array_equal(a[q], b[q]) for q in b)


# ---------------------------------------------------------------------------------------
# appending evaluations at fixed hyper-parameters: O(N^2) update of the factor
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("kid,n0,mean", [(wl.SE, 250, "const"), (wl.RQ, 383, "linear"), (wl.SE, 128, "const")])
def test_append_point_equals_fit_from_scratch(gp_mod, kid, n0, mean):
    """`GpRegressor.add_point` (gpmi_append_point: a new row of L by one triangular sweep + fresh alpha) against a
    from-scratch fit on the enlarged data, point after point across a tile boundary (n0 = 383 -> 384 -> 385, 128 ->
    129): L, alpha, LML, predictions; a LinearMean (centred on the data: every prior mean moves with each point)."""
    d, extra = 3, 5
    x, y, e = wl.synthetic_dataset(90 + n0, n0 + extra, d)
    mean_cls = gp_mod.LinearMean if mean == "linear" else gp_mod.ConstantMean
    th_cov = wl.timing_theta(kid, y, d)[1:]
    th = np.concatenate([[y.mean()] + ([0.1, -0.2, 0.05] if mean == "linear" else []), th_cov])
    kw = dict(kernel=kernel_cls(gp_mod, kid), mean=mean_cls)
    gp = gp_mod.GpRegressor(x[:n0], y[:n0], y_err=e[:n0], hyperpars=th, reserve=extra, **kw)
    pts = wl.query_points(91, 40, d)
    for k in range(extra):
        n = n0 + k
        gp.add_point(x[n], y[n], e[n])
        ref = gp_mod.GpRegressor(x[: n + 1], y[: n + 1], y_err=e[: n + 1], hyperpars=th, **kw)
        check_each(gp.alpha, ref.alpha, 1e-11, what="alpha after append")
        check(gp._logdet, ref._logdet, 1e-12, what="log-determinant after append")
        mu, sig = gp(pts)
        rmu, rsig = ref(pts)
        check(mu, rmu, 1e-11, what="mu after append")
        check(sig, rsig, 1e-11, what="sigma after append")
    check(gp.L, ref.L, 1e-11, what="L after appends")
    check(gp.marginal_likelihood(th), ref.marginal_likelihood(th), 1e-12, what="LML on the appended data")
    lm, ls = gp.loo_predictions()
    rm, rs = ref.loo_predictions()
    check(lm, rm, 1e-10, what="loo mean after appends")
    # capacity exhausted (reserve rounded up to the tile): the next append re-fits from scratch, same result
    gp2 = gp_mod.GpRegressor(x[:128], y[:128], y_err=e[:128], hyperpars=th, **kw)  # no reserve, n = capacity
    gp2.add_point(x[128], y[128], e[128])
    ref2 = gp_mod.GpRegressor(x[:129], y[:129], y_err=e[:129], hyperpars=th, **kw)
    check_each(gp2.alpha, ref2.alpha, 1e-11, what="alpha after a re-fitting append")


def test_failed_append_leaves_the_model_untouched(gp_mod):
    """A point that makes the factor update fail (pivot <= 0, e.g. a duplicate without an error bar at fixed
    hyper-parameters; forced here by making the device call report it): `add_point` raises LinAlgError and the host
    state is as before - n_points, x, y, alpha, predictions - as is GpOptimiser's own record; a following valid append
    works.  HeteroscedasticNoise models are refused up front."""
    from numpy.linalg import LinAlgError

    x, y, _ = wl.synthetic_dataset(31, 200, 2)
    e = np.full(200, 0.1)
    th = np.array([y.mean(), np.log(y.std()), np.log(0.4), np.log(0.4)])
    gp = gp_mod.GpRegressor(x[:150], y[:150], y_err=e[:150], hyperpars=th, reserve=64)
    pts = wl.query_points(31, 20, 2)
    mu0, sig0 = gp(pts)
    alpha0 = gp.alpha.copy()
    real = gp.engine.append_point
    gp.engine.append_point = lambda *a: (None, 0.0, 151)  # the device reports pivot 151 <= 0 and has written nothing
    with pytest.raises(LinAlgError):
        gp.add_point(x[7], y[7], 0.0)
    gp.engine.append_point = real
    assert gp.n_points == 150 and gp.x.shape == (150, 2) and gp.y.size == 150
    assert np.array_equal(gp.alpha, alpha0)
    mu1, sig1 = gp(pts)
    assert np.array_equal(mu1, mu0) and np.array_equal(sig1, sig0)
    gp.add_point(x[150], y[150], e[150])
    ref = gp_mod.GpRegressor(x[:151], y[:151], y_err=e[:151], hyperpars=th)
    check_each(gp.alpha, ref.alpha, 1e-11, what="alpha after a failed and a valid append")
    het = gp_mod.GpRegressor(x[:40], y[:40], kernel=gp_mod.SquaredExponential() + gp_mod.HeteroscedasticNoise())
    with pytest.raises(NotImplementedError):
        het.add_point(x[41], y[41])


def test_gp_optimiser_reusing_hyperparameters(gp_mod):
    """GpOptimiser(reuse_hyperpars=True): add_evaluation appends in O(N^2) and the model equals a fresh regressor on
    all evaluations at the same hyper-parameters."""
    bx, by, bounds = _bo_problem()
    th = np.array([by.mean(), np.log(by.std()), np.log(2.0), np.log(2.0)])
    opt = gp_mod.GpOptimiser(bx, by, bounds=bounds, hyperpars=th, reuse_hyperpars=True)
    np.random.seed(2)
    for _ in range(3):
        nx = opt.propose_evaluation()
        opt.add_evaluation(nx, float(np.sin(nx[0]) + 0.1 * nx[1]))
    assert opt.y.size == by.size + 3 and np.array_equal(opt.gp.hyperpars, th)
    ref = gp_mod.GpRegressor(opt.x, opt.y, hyperpars=th)
    check_each(opt.gp.alpha, ref.alpha, 1e-9, what="alpha of the appended optimiser model")  # y_err = None: cond(K) ~ 1e10


def test_change_point_loo_gradient_batch(golden, gp_mod):
    """Round 5: the leave-one-out objective and gradient of two-region ChangePoint models (+ WhiteNoise) for a batch of
    hyper-parameter vectors in lockstep (gpmi_loo_grad_batch_mix) against the reference (tests/golden/cpx.npz:
    regression.py:489-526 with covariance.py:529-594) and against one-at-a-time evaluations through the dense path."""
    g = golden("cpx")
    x, y, e = g["x"], g["y"], g["y_err"]
    rng = np.random.default_rng(13)
    for tag, wn in (("sese", False), ("sesewn", True)):
        cov = gp_mod.ChangePoint(kernels=[gp_mod.SquaredExponential, gp_mod.SquaredExponential])
        if wn:
            cov = cov + gp_mod.WhiteNoise()
        th = g[f"{tag}_theta"]
        gp2 = gp_mod.GpRegressor(x, y, y_err=e, kernel=cov, hyperpars=th)
        thetas = np.vstack([th] + [th + 0.05 * rng.standard_normal(th.size) for _ in range(3)])
        vals, grads = gp2.loo_likelihood_gradient_batch(thetas)
        check(vals[0], g[f"{tag}_loo"], what="2-region loo, lockstep batch")
        check_each(grads[0], g[f"{tag}_loo_grad"], what="2-region loo gradient, lockstep batch")
        for t, v, gr in zip(thetas[1:], vals[1:], grads[1:]):
            a, b = gp2._dense_loo_gradient(t)  # dense device path, one at a time
            check(v, a, what="2-region loo: batch against single")
            check_each(gr, b, what="2-region loo gradient: batch against single")
        assert gp_mod.GpRegressor(x, y, y_err=e, kernel=cov, hyperpars=th, cross_val=True)._lockstep_search()


def test_heteroscedastic_loo_gradient_batch(golden, gp_mod):
    """Round 5: the leave-one-out objective and gradient of SE + HeteroscedasticNoise for a batch of hyper-parameter vectors in
    lockstep (gpmi_loo_grad_batch_noise: every evaluation its own noise variances in the K-build, diag(K^-1 diag(c2) K^-1) back
    for the per-point components) - against the reference's value and 99-component gradient (tests/golden/cpx.npz,
    regression.py:489-526 with covariance.py:683-689) and against one-at-a-time evaluations through the dense path."""
    g = golden("cpx")
    xh, yh, eh = wl.synthetic_dataset(77, 96, 1)
    thh = g["het_theta"]
    gph = gp_mod.GpRegressor(xh, yh, y_err=eh, kernel=gp_mod.SquaredExponential() + gp_mod.HeteroscedasticNoise(), hyperpars=thh)
    rng = np.random.default_rng(12)
    thetas = np.vstack([thh] + [thh + 0.1 * rng.standard_normal(thh.size) for _ in range(4)])
    vals, grads = gph.loo_likelihood_gradient_batch(thetas)
    check(vals[0], g["het_loo"], what="heteroscedastic loo, lockstep batch")
    check_each(grads[0], g["het_loo_grad"], what="heteroscedastic loo gradient, lockstep batch (99 components)")
    for t, v, gr in zip(thetas[1:], vals[1:], grads[1:]):
        a, b = gph._dense_loo_gradient(t)  # dense device path, one at a time
        check(v, a, what="heteroscedastic loo: batch against single")
        check_each(gr, b, what="heteroscedastic loo gradient: batch against single")
    # a batch of one takes the same kernels
    v1, g1 = gph.loo_likelihood_gradient_batch(thetas[2:3])
    check(v1[0], vals[2], 1e-12, what="heteroscedastic loo: batch of one")
    check(g1[0], grads[2], 1e-11, what="heteroscedastic loo gradient: batch of one")
    # and the cross-validated search of such a model runs its starts in lockstep on it
    assert gp_mod.GpRegressor(xh, yh, y_err=eh, kernel=gp_mod.SquaredExponential() + gp_mod.HeteroscedasticNoise(),
                              hyperpars=thh, cross_val=True)._lockstep_search()


# ---------------------------------------------------------------------------------------
# gradients without a fused device kernel: dense device path + the objects' own derivative matrices
# ---------------------------------------------------------------------------------------
def test_three_region_change_point_and_loo_gradients_vs_reference(golden, gp_mod):
    """ChangePoint over three regions (LML and LOO gradients), LOO gradients of two-region ChangePoint (+ WhiteNoise)
    and of SE + HeteroscedasticNoise against the reference (tests/golden/cpx.npz) - through the fused mixture / per-point
    noise kernels (round 5: any number of regions; single evaluations are lockstep batches of one) AND through the dense
    device path they replaced (K^-1, alpha, p, W from gpmi_lml_dense / gpmi_loo_dense, the contraction with each
    component's own dK on the host), which still serves sizes beyond the lockstep limit."""
    g = golden("cpx")
    x, y, e = g["x"], g["y"], g["y_err"]
    cov3 = gp_mod.ChangePoint(kernels=[gp_mod.SquaredExponential, gp_mod.SquaredExponential, gp_mod.RationalQuadratic])
    th3 = g["cp3_thetas"]
    gp = gp_mod.GpRegressor(x, y, y_err=e, kernel=cov3, hyperpars=th3[0])
    assert list(g["cp3_labels"]) == gp.hyperpar_labels
    for t, v, gr, lv, lg in zip(th3, g["cp3_lml"], g["cp3_grad"], g["cp3_loo"], g["cp3_loo_grad"]):
        a, b = gp.marginal_likelihood_gradient(t)  # (round 5: the fused mixture path with the caller's row-sum weights)
        check(a, v, what="3-region lml")
        check_each(b, gr, what="3-region lml gradient")
        a, b = gp.loo_likelihood_gradient(t)
        check(a, lv, what="3-region loo")
        check_each(b, lg, what="3-region loo gradient")
    # round 5: three regions in lockstep batches too (gpmi_lml_grad_batch_mix / gpmi_loo_grad_batch_mix with hw: the
    # reference differentiates a change-point through ONE window factor of each neighbouring sub-kernel, covariance.py:588-593)
    rng = np.random.default_rng(14)
    more = np.vstack([th3, th3[0] + 0.008 * rng.standard_normal((2, th3.shape[1]))])  # (the window widths are 0.04 - 0.05)
    for b_ in (len(more), 1):
        f, gr = gp.marginal_likelihood_gradient_batch(more[:b_])
        lf, lgr = gp.loo_likelihood_gradient_batch(more[:b_])
        for k in range(min(b_, len(th3))):
            check(f[k], g["cp3_lml"][k], what=f"3-region lml, lockstep batch of {b_}")
            check_each(gr[k], g["cp3_grad"][k], what=f"3-region lml gradient, lockstep batch of {b_}")
            check(lf[k], g["cp3_loo"][k], what=f"3-region loo, lockstep batch of {b_}")
            check_each(lgr[k], g["cp3_loo_grad"][k], what=f"3-region loo gradient, lockstep batch of {b_}")
        for k in range(len(th3), b_):
            a, b = gp.marginal_likelihood_gradient(more[k])
            check(f[k], a, 1e-12, "3-region lml: batch against single")
            check_each(gr[k], b, 1e-11, what="3-region lml gradient: batch against single")
            a, b = gp._dense_loo_gradient(more[k])  # dense device path, one at a time
            check(lf[k], a, what="3-region loo: batch against single")
            check_each(lgr[k], b, what="3-region loo gradient: batch against single")
    assert gp._lockstep_search()
    assert gp_mod.GpRegressor(x, y, y_err=e, kernel=cov3, hyperpars=th3[0], cross_val=True)._lockstep_search()
    for tag, wn in (("sese", False), ("sesewn", True)):
        cov = gp_mod.ChangePoint(kernels=[gp_mod.SquaredExponential, gp_mod.SquaredExponential])
        if wn:
            cov = cov + gp_mod.WhiteNoise()
        th = g[f"{tag}_theta"]
        gp2 = gp_mod.GpRegressor(x, y, y_err=e, kernel=cov, hyperpars=th)
        for fn, how in ((gp2.loo_likelihood_gradient, "lockstep batch of one"), (gp2._dense_loo_gradient, "dense path")):
            a, b = fn(th)
            check(a, g[f"{tag}_loo"], what=f"2-region loo ({how})")
            check_each(b, g[f"{tag}_loo_grad"], what=f"2-region loo gradient ({how})")
    a, b = gp._dense_loo_gradient(th3[0])
    check(a, g["cp3_loo"][0], what="3-region loo (dense path)")
    check_each(b, g["cp3_loo_grad"][0], what="3-region loo gradient (dense path)")
    a, b = gp._dense_lml_gradient(th3[0])
    check(a, g["cp3_lml"][0], what="3-region lml (dense path)")
    check_each(b, g["cp3_grad"][0], what="3-region lml gradient (dense path)")
    xh, yh, eh = wl.synthetic_dataset(77, 96, 1)
    thh = g["het_theta"]
    gph = gp_mod.GpRegressor(xh, yh, y_err=eh, kernel=gp_mod.SquaredExponential() + gp_mod.HeteroscedasticNoise(), hyperpars=thh)
    for fn, how in ((gph.loo_likelihood_gradient, "lockstep batch of one"), (gph._dense_loo_gradient, "dense path")):
        a, b = fn(thh)
        check(a, g["het_loo"], what=f"heteroscedastic loo ({how})")
        check_each(b, g["het_loo_grad"], what=f"heteroscedastic loo gradient, 99 components ({how})")
    # cross_val=True search now works for these kernels too
    np.random.seed(6)
    gcv = gp_mod.GpRegressor(xh, yh, y_err=eh, kernel=gp_mod.SquaredExponential() + gp_mod.WhiteNoise(), cross_val=True, n_starts=2)
    assert np.isfinite(gcv.loo_likelihood(gcv.hyperpars))


def test_four_region_change_point_vs_reference(golden, gp_mod):
    """ChangePoint over FOUR regions [SE, RQ, SE, SE] + WhiteNoise - as many sub-kernels as the fused mixture path carries -
    against the reference (tests/golden/cp4.npz: covariance.py:529-594 in regression.py:218-244, 188-216, 489-567): fit,
    prediction, LML and leave-one-out values with their 17-component gradients, one evaluation at a time and in lockstep
    batches (the window row sums with the caller's weights, include/gpmi.h: hw)."""
    g = golden("cp4")
    x, y, e, pts, th = g["x"], g["y"], g["y_err"], g["pts"], g["thetas"]
    cov = gp_mod.ChangePoint(kernels=[gp_mod.SquaredExponential, gp_mod.RationalQuadratic, gp_mod.SquaredExponential,
                                      gp_mod.SquaredExponential]) + gp_mod.WhiteNoise()
    gp = gp_mod.GpRegressor(x, y, y_err=e, kernel=cov, hyperpars=th[0])
    assert list(g["labels"]) == gp.hyperpar_labels and gp._mix is not None and gp._mix.n_kernels == 4
    check_each(gp.alpha, g["alpha"], what="4-region alpha")
    mu, sig = gp(pts)
    check(mu, g["mu"], what="4-region mu")
    check(sig, g["sig"], what="4-region sig")
    check([gp.marginal_likelihood(t) for t in th], g["lml"], what="4-region lml")
    for t, v, gr, lv, lg in zip(th, g["lml2"], g["grad"], g["loo"], g["loo_grad"]):
        a, b = gp.marginal_likelihood_gradient(t)
        check(a, v, what="4-region lml (gradient call)")
        check_each(b, gr, what="4-region lml gradient")
        a, b = gp.loo_likelihood_gradient(t)
        check(a, lv, what="4-region loo")
        check_each(b, lg, what="4-region loo gradient")
    f, gr = gp.marginal_likelihood_gradient_batch(th)
    lf, lgr = gp.loo_likelihood_gradient_batch(th)
    check(f, g["lml2"], what="4-region lml, lockstep batch")
    check(lf, g["loo"], what="4-region loo, lockstep batch")
    for k in range(len(th)):
        check_each(gr[k], g["grad"][k], what="4-region lml gradient, lockstep batch")
        check_each(lgr[k], g["loo_grad"][k], what="4-region loo gradient, lockstep batch")
    a, b = gp._dense_lml_gradient(th[1])  # the path these replaced, still serving sizes beyond the lockstep limit
    check(a, g["lml2"][1], what="4-region lml (dense path)")
    check_each(b, g["grad"][1], what="4-region lml gradient (dense path)")
    assert gp._lockstep_search()
    mu2, _ = gp(pts)  # the likelihood evaluations used the shared weight buffers: the fit is restored lazily
    check(mu2, g["mu"], what="4-region mu after the likelihood evaluations")


def test_kernel_call_reuses_its_device_context(gp_mod):
    """`cov(u, v, theta)` keeps one device context per point set instead of creating one per call."""
    x, y, e = wl.synthetic_dataset(3, 200, 2)
    cov = gp_mod.SquaredExponential()
    cov.pass_spatial_data(x)
    th = np.array([0.1, -0.3, 0.2])
    u = wl.query_points(3, 7, 2)
    a = cov(u, x, th)
    eng = cov._own_engine()
    b = cov(u, x, th)
    assert cov._own_engine() is eng and np.array_equal(a, b) and getattr(cov, "_cross", None) is None
    other = wl.query_points(4, 50, 2)
    c1 = cov(u, other, th)
    held = cov._cross[1]
    c2 = cov(u[:3], other, th)
    assert cov._cross[1] is held and np.array_equal(c1[:3], c2)
    from oracle import gp_oracle as orc

    check(c1, orc.kernel_cross(wl.SE, u, other, th), 1e-13, "cross-covariance against another point set")


@pytest.mark.gpu
@pytest.mark.parametrize("kid", [wl.SE, wl.RQ])
def test_covariance_at_extreme_hyperparameters_vs_oracle(gp_mod, kid):
    """The K-build's own exp / log1p (csrc/kmath.h) where the library routines they replace are exercised least:
    length scales from e^-6 to e^+6 (arguments of exp from 0 to far below the underflow threshold), RationalQuadratic
    kappa from e^-5 to e^+8 (log1p of 1e-12 .. 1e+9), points that coincide (s = 0 exactly) - against the oracle's NumPy
    values element by element: an entry is held to 4 ulp of the amplitude plus 1e-13 of itself (plus, for the
    RationalQuadratic kernel, the 4 kappa ulp that the REFERENCE's `(1 + z / kappa) ** -kappa` loses to the rounding of its
    base - the device form exp(-kappa log1p(z / kappa)) does not).
    Reference: covariance.py:247-255 (SquaredExponential.__call__), :343-348 (RationalQuadratic.__call__)."""
    from oracle import gp_oracle as orc

    rng = np.random.default_rng(17)
    d = 3
    u = rng.uniform(-1, 1, (150, d))
    v = np.concatenate([rng.uniform(-1, 1, (200, d)), u[:20]])  # (coincident points included)
    cov = kernel_cls(gp_mod, kid)()
    cov.pass_spatial_data(v)
    worst = 0.0
    for log_a in (-3.0, 0.0, 4.0):
        for log_l in (-6.0, -2.5, 0.0, 2.5, 6.0):
            for log_k in ((None,) if kid == wl.SE else (-5.0, -1.0, 0.0, 3.0, 8.0)):
                head = [log_a] if kid == wl.SE else [log_a, log_k]
                th = np.array(head + [log_l + 0.1 * i for i in range(d)])
                got = cov(u, v, th)
                ref = orc.kernel_cross(kid, u, v, th)
                a2 = np.exp(2 * log_a)
                eps = np.finfo(float).eps
                rel_ref = 1e-13 + (0.0 if kid == wl.SE else 4 * eps * np.exp(log_k))
                err = np.abs(got - ref) / (4 * eps * a2 + rel_ref * np.abs(ref))
                worst = max(worst, float(err.max()))
                assert np.isfinite(got).all() and (got >= 0).all() and (got <= a2 * (1 + 1e-15)).all(), th
    _record("K entries at extreme hyper-parameters (units of the stated bound)", worst * 1e-13, 1e-13)
    assert worst <= 1.0, worst
