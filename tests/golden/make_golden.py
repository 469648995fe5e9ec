"""
Generate golden vectors for the GP hot path by IMPORTING the reference
(C-bowman/inference-tools, mounted read-only at /root/reference) and running
it on the seeded synthetic inputs of `workloads.py`.

Run in the build container only:   python tests/golden/make_golden.py [case ...]

The reference's source never travels: this script only imports it (with
PYTHONDONTWRITEBYTECODE so nothing is written into the reference tree), and the
outputs are stored as small .npz fixtures beside this file.  On a machine
without /root/reference (the GPU box) the script exits with a message.

Cases
  t32        the reference's own GP test data (tests/gp/test_GpRegressor.py:36-42), SE and RQ,
             full matrices + every public method of the path
  cfg1       BASELINE config 1: SE, N=512, d=2
  rq256      RQ, N=256, d=16 (config-3 shape at a size the reference classes can hold)
  cfg4       BASELINE config 4: SE, N=4096, d=4, 1000 EI candidates
  cfg2       BASELINE config 2: SE, N=8192, d=8 (needs ~16 GB RSS, several minutes)
  fail       a theta for which numpy.linalg.cholesky raises (pins the -1e50 path)
  cp         ChangePoint over [SE, SE], [SE, RQ], [SE, SE] + WhiteNoise, N=200 d=2
  het        SE + HeteroscedasticNoise, N=96: LML, gradient (99 parameters), fit + predict
  linv       GpLinearInverter: 1-D deconvolution (32 x 64) and 2-D tomography (300 x 400), SE / RQ / SE+WhiteNoise
  search     the stochastic callers under numpy.random.seed: multistart_bfgs (start positions, theta*, LML(theta*); LML and
             LOO objectives), differential_evo, AcquisitionFunction.starting_positions, GpOptimiser.propose_evaluation
  cpx        gradients beyond the two-region LML case: ChangePoint over THREE regions (LML and LOO gradients), LOO gradients of
             two-region ChangePoint (+ WhiteNoise) and of SE + HeteroscedasticNoise (until round 5 served by the dense device
             path only; since then also by the fused mixture / per-point-noise kernels)
  cp4        ChangePoint over FOUR regions [SE, RQ, SE, SE] + WhiteNoise, N=180 d=1: fit, predict, LML and LOO gradients
  plugin     a user-defined covariance function written against the plugin ABC only (Matern-3/2, workloads.Matern32Math):
             every public GpRegressor method, the seeded hyper-parameter search and an EI proposal
  linvp      GpLinearInverter with a prior that has no device kernel: the plugin Matern-3/2 (ABC only) on the tomography
             problem and ChangePoint over [SE, RQ] on the deconvolution problem - LML, gradient, posterior
  means      LinearMean / QuadraticMean: labels, bounds, fit, predict, LML, LML gradient (mean-parameter components
             included), LOO gradient, posterior
  head16k    the metric's own size: SE, N=16384, d=8 (fit at the timing theta: alpha at 64 indices, |alpha|, diag(L),
             logdet, LML at 3 thetas, mu / sigma at 64 query points, posterior at 16).  The reference's classes need
             3 x N^2 d x 8 B = 52 GB there (covariance.py:218-219,254), more than this container can give them, so
             this case and the next are produced by the ROW-CHUNKED ORACLE (oracle/gp_oracle.py) - the same
             element-wise operations in the same order, the same numpy.linalg.cholesky / solve_triangular - which
             tests/test_oracle_golden.py pins to the imported reference up to N = 8192 (K bit-exact).  The script
             first re-checks that pin at N = 2048 with these very code paths before it writes anything.
  gradpin    the reference's LML / LOO gradients at SE N=2048 d=8 and RQ N=1536 d=16: pins of the oracle's
             one-matrix-at-a-time gradients
  cfg2g      BASELINE config 2 (SE, N=8192, d=8): the reference's LML gradient (2 thetas), LOO value, gradient and
             predictions (~20 GB RSS)
  head16kg   SE, N=16384, d=8: LML gradient (2 thetas) and LOO gradient from the oracle's one-matrix-at-a-time forms
  cfg3_16k   BASELINE config 3 at full size: RQ, N=16384, d=16, 8 thetas of the 64-point grid (one GPU's share of
             the 8-GPU sweep: grid rows 0, 9, 18, ... 63): LML and logdet each; alpha / predict at the first
"""
import os
import sys
import types
import warnings

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

REF = "/root/reference"
if not os.path.isdir(os.path.join(REF, "inference")):
    print("reference tree not present - golden vectors can only be generated in the build container")
    sys.exit(0)

# `import inference` needs setuptools_scm (inference/__init__.py:3-8), which is absent:
# pre-register a bare package pointing at the reference source instead.
pkg = types.ModuleType("inference")
pkg.__path__ = [os.path.join(REF, "inference")]
sys.modules["inference"] = pkg

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

import numpy as np  # noqa: E402
from inference.gp import (  # noqa: E402
    GpRegressor,
    SquaredExponential,
    RationalQuadratic,
    WhiteNoise,
    HeteroscedasticNoise,
    ChangePoint,
    ExpectedImprovement,
    UpperConfidenceBound,
    MaxVariance,
)
import workloads as wl  # noqa: E402

IDX64 = None


def idx64(n):
    return np.unique(np.linspace(0, n - 1, 64).astype(int))


def kernel_cls(kid):
    return SquaredExponential if kid == wl.SE else RationalQuadratic


def common_outputs(gp, thetas, pts, out, prefix="", full=False, grads=True):
    n = gp.n_points
    ii = idx64(n)
    out[prefix + "hp_bounds"] = np.array(gp.hp_bounds, dtype=float)
    out[prefix + "labels"] = np.array(gp.hyperpar_labels)
    out[prefix + "str"] = np.array(str(gp))
    out[prefix + "thetas"] = thetas
    out[prefix + "lml"] = np.array([gp.marginal_likelihood(t) for t in thetas])
    if grads:
        res = [gp.marginal_likelihood_gradient(t) for t in thetas]
        out[prefix + "lml_g_val"] = np.array([r[0] for r in res])
        out[prefix + "lml_g_grad"] = np.array([r[1] for r in res])
    # fitted state at thetas[0]
    gp.set_hyperparameters(thetas[0])
    out[prefix + "alpha_norm"] = np.linalg.norm(gp.alpha)
    out[prefix + "alpha_idx"] = ii
    out[prefix + "alpha_sub"] = gp.alpha[ii]
    out[prefix + "diagL_sub"] = np.diagonal(gp.L)[ii]
    out[prefix + "logdet"] = np.log(np.diagonal(gp.L)).sum()
    mu, sig = gp(pts)
    out[prefix + "pts"] = pts
    out[prefix + "mu"] = mu
    out[prefix + "sig"] = sig
    pm, pc = gp.build_posterior(pts[:16])
    out[prefix + "post_mu"] = pm
    out[prefix + "post_cov"] = pc
    if full:
        out[prefix + "K_xx"] = gp.K_xx
        out[prefix + "L"] = gp.L
        out[prefix + "alpha"] = gp.alpha


def case_t32():
    """The reference's own test data: tests/gp/test_GpRegressor.py:36-42."""
    out = {}
    n = 32
    rng = np.random.default_rng(1)
    points = rng.uniform(low=0.0, high=2.0, size=(n, 2))
    values = np.sin(points[:, 0]) * np.cos(points[:, 1]) + rng.normal(scale=0.1, size=n)
    errors = np.full(n, fill_value=0.1)
    out["x"], out["y"], out["y_err"] = points, values, errors
    qrng = np.random.default_rng(99)
    pts = qrng.uniform(0.0, 2.0, size=(24, 2))

    # thetas as in test_marginal_likelihood_gradient (tests/gp/test_GpRegressor.py:65-69)
    trng = np.random.default_rng(123)
    th_se = trng.uniform(low=[-0.3, -1.5, 0.1, 0.1], high=[0.3, 0.5, 1.5, 1.5], size=[8, 4])
    th_rq = trng.uniform(
        low=[-0.3, -1.5, -1.0, 0.1, 0.1], high=[0.3, 0.5, 3.0, 1.5, 1.5], size=[8, 5]
    )
    for name, kid, th in (("se_", wl.SE, th_se), ("rq_", wl.RQ, th_rq)):
        gp = GpRegressor(points, values, y_err=errors, hyperpars=th[0], kernel=kernel_cls(kid))
        common_outputs(gp, th, pts, out, prefix=name, full=True)
        # kernel-level outputs
        K, dK = gp.cov.covariance_and_gradients(th[1][1:])
        out[name + "cov_K"] = K
        out[name + "cov_dK"] = np.array(dK)
        out[name + "cov_cross"] = gp.cov(pts, points, th[1][1:])
        # LOO trio (regression.py:451-526)
        gp.set_hyperparameters(th[0])
        lm, ls = gp.loo_predictions()
        out[name + "loo_mu"], out[name + "loo_sig"] = lm, ls
        out[name + "loo"] = np.array([gp.loo_likelihood(t) for t in th])
        res = [gp.loo_likelihood_gradient(t) for t in th]
        out[name + "loo_g_val"] = np.array([r[0] for r in res])
        out[name + "loo_g_grad"] = np.array([r[1] for r in res])
    # SE-only spatial gradients (RQ raises NotImplementedError: covariance.py:38-44)
    gp = GpRegressor(points, values, y_err=errors, hyperpars=th_se[0])
    g_mu, g_cov = gp.gradient(pts[:16])
    s_mu, s_var = gp.spatial_derivatives(pts[:16])
    out["se_grad_mu"], out["se_grad_cov"] = g_mu, g_cov
    out["se_sd_mu"], out["se_sd_var"] = s_mu, s_var
    # acquisition functions on the fitted SE GP (acquisition.py:44-232)
    for nm, acq in (("ei", ExpectedImprovement()), ("ucb", UpperConfidenceBound()), ("mv", MaxVariance())):
        acq.update_gp(gp)
        out[f"se_{nm}_call"] = np.array([acq(p) for p in pts])
        out[f"se_{nm}_opt"] = np.array([acq.opt_func(p) for p in pts])
        r = [acq.opt_func_gradient(p) for p in pts]
        out[f"se_{nm}_optg_val"] = np.array([float(np.squeeze(a)) for a, _ in r])
        out[f"se_{nm}_optg_grad"] = np.array([b for _, b in r])
        out[f"se_{nm}_conv"] = np.array([acq.convergence_metric(p) for p in pts[:4]])
    # composite kernel SE + WhiteNoise (covariance.py:47-105,108-178)
    thc = np.array([0.1, -0.4, 0.3, 0.6, -2.0])
    gpc = GpRegressor(points, values, y_err=errors, hyperpars=thc, kernel=SquaredExponential() + WhiteNoise())
    out["sewn_theta"] = thc
    out["sewn_labels"] = np.array(gpc.hyperpar_labels)
    out["sewn_hp_bounds"] = np.array(gpc.hp_bounds, dtype=float)
    out["sewn_lml"] = gpc.marginal_likelihood(thc)
    v, g = gpc.marginal_likelihood_gradient(thc)
    out["sewn_lml_g_val"], out["sewn_lml_g_grad"] = v, g
    out["sewn_alpha"] = gpc.alpha
    m, s = gpc(pts)
    out["sewn_mu"], out["sewn_sig"] = m, s
    # 1-D data (regression.py:110-112 reshape path) as in tests/gp/test_GpRegressor.py:97-117
    rng1 = np.random.default_rng(42)
    N, S = 10, 1.1
    x1 = np.linspace(0, 10, N)
    y1 = 0.3 * x1 + 0.02 * x1**3 + 5.0 + rng1.normal(size=N) * S
    e1 = np.zeros(N) + S
    th1 = np.array([y1.mean(), np.log(y1.std()), np.log(2.0)])
    gp1 = GpRegressor(x1, y1, y_err=e1, hyperpars=th1)
    sx = np.linspace(0, 10, 30)
    out["d1_x"], out["d1_y"], out["d1_err"], out["d1_theta"], out["d1_pts"] = x1, y1, e1, th1, sx
    out["d1_mu"], out["d1_sig"] = gp1(sx)
    out["d1_grad_mu"], out["d1_grad_cov"] = gp1.gradient(sx)
    out["d1_sd_mu"], out["d1_sd_var"] = gp1.spatial_derivatives(sx)
    out["d1_hp_bounds"] = np.array(gp1.hp_bounds, dtype=float)
    return out


def case_synthetic(cfg, kid, n, d, n_theta, m, grads=True):
    out = {}
    x, y, y_err = wl.synthetic_dataset(cfg, n, d)
    thetas = wl.theta_set(kid, y, d, n_theta)
    pts = wl.query_points(cfg, m, d)
    gp = GpRegressor(x, y, y_err=y_err, hyperpars=thetas[0], kernel=kernel_cls(kid))
    common_outputs(gp, thetas, pts, out, grads=grads)
    out["meta"] = np.array([cfg, kid, n, d])
    return out, gp


def case_cfg1():
    out, gp = case_synthetic(1, wl.SE, 512, 2, 8, 64)
    return out


def case_rq256():
    out, gp = case_synthetic(3, wl.RQ, 256, 16, 8, 64)
    return out


def case_cfg4():
    out, gp = case_synthetic(4, wl.SE, 4096, 4, 2, 64, grads=True)
    # 1000 EI candidates (acquisition.py:76-125)
    cand = wl.query_points(4004, 1000, 4)
    ei = ExpectedImprovement()
    ei.update_gp(gp)
    out["cand"] = cand
    out["ei_call"] = np.array([ei(p) for p in cand])
    out["ei_opt"] = np.array([ei.opt_func(p) for p in cand])
    r = [ei.opt_func_gradient(p) for p in cand[:200]]
    out["ei_optg_val"] = np.array([float(np.squeeze(a)) for a, _ in r])
    out["ei_optg_grad"] = np.array([b for _, b in r])
    out["mu_max"] = ei.mu_max
    # force the Z < -3 branch: same GP, a y_max far above the data
    ei.mu_max = gp.y.max() + 2.0
    out["ei_far_mu_max"] = ei.mu_max
    out["ei_far_call"] = np.array([ei(p) for p in cand[:200]])
    out["ei_far_opt"] = np.array([ei.opt_func(p) for p in cand[:200]])
    r = [ei.opt_func_gradient(p) for p in cand[:200]]
    out["ei_far_optg_val"] = np.array([float(np.squeeze(a)) for a, _ in r])
    out["ei_far_optg_grad"] = np.array([b for _, b in r])
    return out


def case_cfg2():
    out, gp = case_synthetic(2, wl.SE, 8192, 8, 2, 64, grads=False)
    return out


def case_fail():
    """Inputs for which numpy.linalg.cholesky raises LinAlgError, so that
    marginal_likelihood returns -1e50 (regression.py:540-542):
    a symmetric but indefinite y_cov (regression.py:262-293 accepts it unchecked),
    (an overflowing amplitude is NOT such a case: cholesky returns non-finite values and
    scipy's solve_triangular then raises ValueError)."""
    out = {}
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 1, (48, 2))
    y = np.sin(3 * x[:, 0]) + x[:, 1]
    th0 = np.array([y.mean(), 0.0, 0.0, 0.0])
    y_cov = np.diag(np.where(np.arange(48) % 7 == 3, -1.5, 0.01))
    gp = GpRegressor(x, y, y_cov=y_cov, hyperpars=np.array([y.mean(), 2.0, -3.0, -3.0]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        v_cov = gp.marginal_likelihood(th0)
        v_ok = gp.marginal_likelihood(np.array([y.mean(), 2.0, -3.0, -3.0]))
    assert v_cov == -1e50 and v_ok != -1e50
    out["x"], out["y"], out["y_cov"] = x, y, y_cov
    out["theta_bad"], out["theta_ok"] = th0, np.array([y.mean(), 2.0, -3.0, -3.0])
    out["lml_bad"], out["lml_ok"] = v_cov, v_ok
    return out


def pt_problem():
    """Small GP whose log-marginal likelihood is the chains' posterior (shared with the tests)."""
    rng = np.random.default_rng(31)
    x = np.sort(rng.uniform(0, 6, 48))
    y = np.sin(x) + 0.3 * np.cos(2.5 * x) + 0.2 * rng.normal(size=48)
    y_err = np.full(48, 0.2)
    start = np.array([y.mean(), np.log(y.std()), np.log(1.0)])
    widths = np.array([0.1, 0.2, 0.2])
    return x, y, y_err, start, widths


def case_pt():
    """Teacher-forced traces of GibbsChain (gibbs.py:627-656) and ParallelTempering
    (parallel.py:190-281): every generator is seeded by assignment after construction, the chains'
    posterior is the reference GpRegressor.marginal_likelihood."""
    import random
    from numpy.random import default_rng
    from inference.mcmc import GibbsChain, ParallelTempering

    out = {}
    x, y, y_err, start, widths = pt_problem()
    gp = GpRegressor(x, y, y_err=y_err, hyperpars=start)
    bounds = gp.hp_bounds
    out["hp_bounds"] = np.array(bounds, dtype=float)

    def make_chain(temp, seed, progress=True):
        # display_progress=False makes the reference chain unpicklable (return_chains then fails)
        ch = GibbsChain(posterior=gp.marginal_likelihood, start=start, widths=widths, temperature=temp,
                        display_progress=progress)
        for i, b in enumerate(bounds):
            ch.set_boundaries(i, b)
        ch.rng = default_rng(seed)
        for i, par in enumerate(ch.params):
            par.rng = default_rng(seed + 1 + i)
        return ch

    ch = make_chain(1.0, 100)
    ch.advance(60)
    out["single_samples"] = ch.get_sample(burn=0)
    out["single_probs"] = np.array(ch.probs)
    out["single_sigmas"] = np.array([par.sigma for par in ch.params])

    temps = [1.0, 2.0, 4.0, 8.0]
    chains = [make_chain(t, 1000 + 10 * k) for k, t in enumerate(temps)]
    pt = ParallelTempering(chains)
    pt.rng = default_rng(7)
    random.seed(9)
    pt.advance(40, swap_interval=5)
    got = pt.return_chains()
    pt.shutdown()
    out["temps"] = np.array(temps)
    for k, c in enumerate(got):
        out[f"pt_samples_{k}"] = c.get_sample(burn=0)
        out[f"pt_probs_{k}"] = np.array(c.probs)
    out["pt_successful"] = pt.successful_swaps
    out["pt_attempted"] = pt.attempted_swaps
    return out


def case_cp():
    """ChangePoint (covariance.py:371-606) over [SE, SE], [SE, RQ] and [SE, SE] + WhiteNoise on a 200-point 2-D
    set whose smoothness changes along axis 0: labels, bounds, LML, LML gradient, fit (K_xx, alpha) + predict."""
    n, d = 200, 2
    rng = np.random.default_rng(4242)
    x = rng.uniform(0, 1, (n, d))
    y = np.where(x[:, 0] < 0.5, np.sin(3 * x[:, 0] + x[:, 1]), np.sin(25 * x[:, 0]) * np.cos(9 * x[:, 1])) + 0.05 * rng.normal(size=n)
    e = np.full(n, 0.05)
    pts = rng.uniform(0, 1, (50, d))
    out = {"x": x, "y": y, "y_err": e, "pts": pts}
    for tag, subs, wn in (("sese", (wl.SE, wl.SE), False), ("serq", (wl.SE, wl.RQ), False), ("sesewn", (wl.SE, wl.SE), True)):
        cov = ChangePoint(kernels=[kernel_cls(k) for k in subs])
        if wn:
            cov = cov + WhiteNoise()
        thetas = []
        for k in range(3):
            th = [0.1 * k]
            for kid, ell in zip(subs, (0.4, 0.08)):
                th += [-0.2 + 0.1 * k] + ([0.2] if kid == wl.RQ else []) + [np.log(ell) + 0.05 * k, np.log(ell * 2) - 0.05 * k]
            th += [0.45 + 0.03 * k, 0.05 + 0.02 * k]
            if wn:
                th.append(np.log(0.03) + 0.2 * k)
            thetas.append(np.array(th))
        gp = GpRegressor(x, y, y_err=e, kernel=cov, hyperpars=thetas[1])
        out[f"{tag}_thetas"] = np.array(thetas)
        out[f"{tag}_labels"] = np.array(gp.hyperpar_labels)
        out[f"{tag}_bounds"] = np.array(gp.hp_bounds, dtype=float)
        out[f"{tag}_alpha"] = gp.alpha
        out[f"{tag}_K_xx"] = gp.K_xx
        mu, sig = gp(pts)
        out[f"{tag}_mu"], out[f"{tag}_sig"] = mu, sig
        out[f"{tag}_lml"] = np.array([gp.marginal_likelihood(t) for t in thetas])
        res = [gp.marginal_likelihood_gradient(t) for t in thetas]
        out[f"{tag}_lml2"] = np.array([r[0] for r in res])
        out[f"{tag}_grad"] = np.array([r[1] for r in res])
        loo_mu, loo_sig = gp.loo_predictions()
        out[f"{tag}_loo_mu"], out[f"{tag}_loo_sig"] = loo_mu, loo_sig
        pm, pc = gp.build_posterior(pts[:20])
        out[f"{tag}_post_mu"], out[f"{tag}_post_cov"] = pm, pc
        out[f"{tag}_loo"] = np.array([gp.loo_likelihood(t) for t in thetas])
    return out


def case_het():
    """SquaredExponential + HeteroscedasticNoise (covariance.py:608-690) on a 96-point 1-D set (the
    reference's HeteroscedasticNoise.__call__ sizes its zero block by u.size, covariance.py:671-672, so its
    predictions only work for d = 1): 99 hyper-parameters, LML and its gradient, fit + predict."""
    n, d = 96, 1
    x, y, e = wl.synthetic_dataset(77, n, d)
    rng = np.random.default_rng(770)
    out = {}
    for tag, with_err in (("err", True), ("noerr", False)):
        gp = GpRegressor(x, y, y_err=e if with_err else None, kernel=SquaredExponential() + HeteroscedasticNoise(),
                         hyperpars=None if False else np.concatenate([wl.timing_theta(wl.SE, y, d), np.log(0.1) + 0.3 * rng.standard_normal(n)]))
        thetas = [np.concatenate([wl.timing_theta(wl.SE, y, d) + 0.1 * k, np.log(0.1) + 0.3 * rng.standard_normal(n)])
                  for k in range(3)]
        out[f"{tag}_thetas"] = np.array(thetas)
        out[f"{tag}_labels"] = np.array(gp.hyperpar_labels)
        out[f"{tag}_bounds"] = np.array(gp.hp_bounds, dtype=float)
        out[f"{tag}_lml"] = np.array([gp.marginal_likelihood(t) for t in thetas])
        res = [gp.marginal_likelihood_gradient(t) for t in thetas]
        out[f"{tag}_lml2"] = np.array([r[0] for r in res])
        out[f"{tag}_grad"] = np.array([r[1] for r in res])
        gp.set_hyperparameters(thetas[1])
        pts = wl.query_points(77, 40, d)
        mu, sig = gp(pts)
        out[f"{tag}_alpha"] = gp.alpha
        out[f"{tag}_K_xx"] = gp.K_xx
        out[f"{tag}_mu"], out[f"{tag}_sig"] = mu, sig
        out[f"{tag}_loo"] = np.array([gp.loo_likelihood(t) for t in thetas])
    return out


def case_linv():
    """GpLinearInverter (inversion.py): LML, LML gradient, posterior mean / covariance."""
    from inference.gp import GpLinearInverter

    out = {}
    for prob in ("deconv", "tomo"):
        pos, A, y, y_err = wl.linv_problem(prob)
        for tag, kid, wn in (("se", wl.SE, False), ("rq", wl.RQ, False), ("sewn", wl.SE, True)):
            cov = kernel_cls(kid)()
            if wn:
                cov = cov + WhiteNoise()
            gli = GpLinearInverter(y=y, y_err=y_err, model_matrix=A, parameter_spatial_positions=pos,
                                   prior_covariance_function=cov)
            out[f"{prob}_{tag}_labels"] = np.array(gli.hyperpar_labels)
            for i, th in enumerate(wl.linv_thetas(prob, kid, wn)):
                key = f"{prob}_{tag}_{i}"
                out[key + "_theta"] = th
                out[key + "_lml"] = np.array(gli.marginal_likelihood(th))
                l2, g = gli.marginal_likelihood_gradient(th)
                out[key + "_lml2"] = np.array(l2)
                out[key + "_grad"] = g
                pm, pc = gli.calculate_posterior(th)
                out[key + "_pmean"] = pm
                out[key + "_pmean_only"] = gli.calculate_posterior_mean(th)
                out[key + "_pcov"] = pc if prob == "deconv" else pc[IDX_TOMO][:, IDX_TOMO]
    return out


def case_linvp():
    """GpLinearInverter with priors that are not SE / RQ (+ WhiteNoise): any CovarianceFunction object is accepted
    (inversion.py:117-127)."""
    from inference.gp import GpLinearInverter
    from inference.gp.covariance import CovarianceFunction

    class Matern32(wl.Matern32Math, CovarianceFunction):
        pass

    out = {}
    pos, A, y, y_err = wl.linv_problem("tomo")
    gli = GpLinearInverter(y=y, y_err=y_err, model_matrix=A, parameter_spatial_positions=pos,
                           prior_covariance_function=Matern32())
    thetas = np.array([[0.2, np.log(0.6), np.log(0.2), np.log(0.25)], [0.0, np.log(1.1), np.log(0.35), np.log(0.15)]])
    out["m32_thetas"] = thetas
    out["m32_labels"] = np.array(gli.hyperpar_labels)
    out["m32_lml"] = np.array([gli.marginal_likelihood(t) for t in thetas])
    res = [gli.marginal_likelihood_gradient(t) for t in thetas]
    out["m32_lml2"], out["m32_grad"] = np.array([r[0] for r in res]), np.array([r[1] for r in res])
    pm, pc = gli.calculate_posterior(thetas[0])
    out["m32_pmean"], out["m32_pcov"] = pm, pc[IDX_TOMO][:, IDX_TOMO]
    out["m32_pmean_only"] = gli.calculate_posterior_mean(thetas[0])
    pos, A, y, y_err = wl.linv_problem("deconv")
    cp = ChangePoint(kernels=[SquaredExponential, RationalQuadratic])
    gli = GpLinearInverter(y=y, y_err=y_err, model_matrix=A, parameter_spatial_positions=pos,
                           prior_covariance_function=cp)
    th = np.zeros(gli.n_hyperpars)
    lab = gli.hyperpar_labels
    rng = np.random.default_rng(99)
    th[:] = 0.1 * rng.standard_normal(th.size)
    for i, name in enumerate(lab):  # a change point inside the domain, a finite width
        if "location" in name.lower():
            th[i] = 0.1
        if "width" in name.lower():
            th[i] = np.log(0.2) if "log" in name.lower() else 0.2
    out["cp_labels"] = np.array(lab)
    out["cp_theta"] = th
    out["cp_lml"] = np.array(gli.marginal_likelihood(th))
    l2, g = gli.marginal_likelihood_gradient(th)
    out["cp_lml2"], out["cp_grad"] = np.array(l2), g
    pm, pc = gli.calculate_posterior(th)
    out["cp_pmean"], out["cp_pcov"] = pm, pc
    return out


def t32_data():
    rng = np.random.default_rng(1)
    points = rng.uniform(low=0.0, high=2.0, size=(32, 2))
    values = np.sin(points[:, 0]) * np.cos(points[:, 1]) + rng.normal(scale=0.1, size=32)
    return points, values, np.full(32, 0.1)


def bo_problem():
    """The 2-D objective of the optimiser tests (shared with tests/test_gpu_parity.py)."""
    def objective(x):
        return np.sin(0.5 * x[0]) * 3 / (2 + 0.5 * (x[1] - 1.0) ** 2) + 0.1 * x[0]

    rng = np.random.default_rng(4)
    bounds = [(-4.0, 6.0), (-3.0, 5.0)]
    x = rng.uniform([b[0] for b in bounds], [b[1] for b in bounds], size=(8, 2))
    x[5] = [7.0, 1.0]  # one evaluation outside the search bounds: starting_positions draws a uniform start for it
    y = np.array([objective(k) for k in x])
    return x, y, bounds


def case_search():
    """The callers that draw random numbers, run under numpy.random.seed (the legacy global generator is what
    regression.py:589-594, acquisition.py:21-35 and SciPy's differential_evolution use)."""
    from inference.gp import GpOptimiser

    out = {}
    x, y, e = t32_data()
    for tag, kw in (("lml", {}), ("loo", {"cross_val": True}), ("rq", {"kernel": RationalQuadratic})):
        log = []
        orig = GpRegressor.launch_bfgs

        def spy(self, x0, _orig=orig, _log=log):
            res = _orig(self, x0)
            _log.append((np.array(x0), np.array(res[0]), float(res[1])))
            return res

        GpRegressor.launch_bfgs = spy
        try:
            np.random.seed(3)
            gp = GpRegressor(x, y, y_err=e, optimizer="bfgs", n_starts=4, **kw)
        finally:
            GpRegressor.launch_bfgs = orig
        out[f"ms_{tag}_starts"] = np.array([l[0] for l in log])
        out[f"ms_{tag}_ends"] = np.array([l[1] for l in log])
        out[f"ms_{tag}_fvals"] = np.array([l[2] for l in log])
        out[f"ms_{tag}_theta"] = np.array(gp.hyperpars)
        out[f"ms_{tag}_best"] = np.array(gp.model_selector(gp.hyperpars))
        out[f"ms_{tag}_bounds"] = np.array(gp.hp_bounds, dtype=float)
    # default number of starts: int(2 sqrt(P)) + 1 = 5 for P = 4
    np.random.seed(8)
    gp = GpRegressor(x, y, y_err=e)
    out["ms_default_theta"] = np.array(gp.hyperpars)
    out["ms_default_best"] = np.array(gp.marginal_likelihood(gp.hyperpars))
    np.random.seed(5)
    gpd = GpRegressor(x, y, y_err=e, optimizer="diffev")
    out["de_theta"] = np.array(gpd.hyperpars)
    out["de_best"] = np.array(gpd.marginal_likelihood(gpd.hyperpars))

    # acquisition starting positions and the optimiser's proposal
    bx, by, bounds = bo_problem()
    th = np.array([by.mean(), np.log(by.std()), np.log(2.0), np.log(2.0)])
    for nm, acq in (("ei", ExpectedImprovement), ("ucb", UpperConfidenceBound), ("mv", MaxVariance)):
        opt = GpOptimiser(bx, by, bounds=bounds, hyperpars=th, acquisition=acq)
        np.random.seed(21)
        out[f"bo_{nm}_starts"] = np.array(opt.acquisition.starting_positions(bounds))
        np.random.seed(22)
        prop = opt.propose_evaluation()
        out[f"bo_{nm}_proposal"] = np.array(prop)
        out[f"bo_{nm}_value"] = np.array(float(np.squeeze(opt.acquisition.opt_func(np.array(prop)))))
        np.random.seed(23)
        prop_de = opt.propose_evaluation(optimizer="diffev")
        out[f"bo_{nm}_de_value"] = np.array(float(np.squeeze(opt.acquisition.opt_func(np.array(prop_de)))))
    out["bo_theta"] = th
    # one full iteration with the hyper-parameter search inside add_evaluation
    np.random.seed(31)
    opt = GpOptimiser(bx, by, bounds=bounds)
    out["bo_fit_theta"] = np.array(opt.gp.hyperpars)
    out["bo_fit_lml"] = np.array(opt.gp.marginal_likelihood(opt.gp.hyperpars))
    return out


def case_cpx():
    out = {}
    n, d = 200, 2
    rng = np.random.default_rng(4242)
    x = rng.uniform(0, 1, (n, d))
    y = np.where(x[:, 0] < 0.5, np.sin(3 * x[:, 0] + x[:, 1]), np.sin(25 * x[:, 0]) * np.cos(9 * x[:, 1])) + 0.05 * rng.normal(size=n)
    e = np.full(n, 0.05)
    out.update(x=x, y=y, y_err=e)
    # three regions
    cov3 = ChangePoint(kernels=[SquaredExponential, SquaredExponential, RationalQuadratic])
    th3 = np.array([[0.05 * k, -0.2, np.log(0.4), np.log(0.8), -0.1, np.log(0.08), np.log(0.2), 0.1, 0.3, np.log(0.3), np.log(0.5),
                     0.33 + 0.02 * k, 0.05, 0.66, 0.04 + 0.01 * k] for k in range(2)])
    gp = GpRegressor(x, y, y_err=e, kernel=cov3, hyperpars=th3[0])
    out["cp3_thetas"] = th3
    out["cp3_labels"] = np.array(gp.hyperpar_labels)
    res = [gp.marginal_likelihood_gradient(t) for t in th3]
    out["cp3_lml"], out["cp3_grad"] = np.array([r[0] for r in res]), np.array([r[1] for r in res])
    res = [gp.loo_likelihood_gradient(t) for t in th3]
    out["cp3_loo"], out["cp3_loo_grad"] = np.array([r[0] for r in res]), np.array([r[1] for r in res])
    # two regions: LOO gradient
    for tag, subs, wn in (("sese", (wl.SE, wl.SE), False), ("sesewn", (wl.SE, wl.SE), True)):
        cov = ChangePoint(kernels=[kernel_cls(k) for k in subs])
        if wn:
            cov = cov + WhiteNoise()
        th = [0.1, -0.1, np.log(0.4), np.log(0.8), -0.2, np.log(0.08), np.log(0.16), 0.48, 0.07]
        if wn:
            th.append(np.log(0.03))
        th = np.array(th)
        gp = GpRegressor(x, y, y_err=e, kernel=cov, hyperpars=th)
        v, g = gp.loo_likelihood_gradient(th)
        out[f"{tag}_theta"], out[f"{tag}_loo"], out[f"{tag}_loo_grad"] = th, np.array(v), g
    # heteroscedastic noise: LOO gradient (N = 96: the reference materialises 96 dense gradient matrices)
    xh, yh, eh = wl.synthetic_dataset(77, 96, 1)
    rngh = np.random.default_rng(771)
    thh = np.concatenate([wl.timing_theta(wl.SE, yh, 1), np.log(0.1) + 0.3 * rngh.standard_normal(96)])
    gp = GpRegressor(xh, yh, y_err=eh, kernel=SquaredExponential() + HeteroscedasticNoise(), hyperpars=thh)
    v, g = gp.loo_likelihood_gradient(thh)
    out["het_theta"], out["het_loo"], out["het_loo_grad"] = thh, np.array(v), g
    return out


def case_cp4():
    """ChangePoint over FOUR regions (the most the fused mixture path carries) + WhiteNoise: fit, prediction, LML and LOO
    with their gradients at two hyper-parameter vectors (round 5: the window row sums with the caller's weights)."""
    out = {}
    n = 180
    rng = np.random.default_rng(4343)
    x = np.sort(rng.uniform(0, 1, n)).reshape(-1, 1)
    xx = x[:, 0]
    y = np.select([xx < 0.25, xx < 0.5, xx < 0.75], [np.sin(6 * xx), 0.5 * np.sin(40 * xx), 0.3 * xx], np.cos(18 * xx))
    y = y + 0.04 * rng.normal(size=n)
    e = np.full(n, 0.04)
    pts = np.linspace(0.02, 0.98, 37).reshape(-1, 1)
    out.update(x=x, y=y, y_err=e, pts=pts)
    cov = ChangePoint(kernels=[SquaredExponential, RationalQuadratic, SquaredExponential, SquaredExponential]) + WhiteNoise()
    th = np.array([[0.02 * k, -0.3, np.log(0.2), -0.5, 0.2, np.log(0.03), -0.9, np.log(0.5), -0.4, np.log(0.06 + 0.01 * k),
                    0.25, 0.02, 0.5 + 0.01 * k, 0.03, 0.75, 0.025, np.log(0.02)] for k in range(2)])
    gp = GpRegressor(x, y, y_err=e, kernel=cov, hyperpars=th[0])
    out["thetas"] = th
    out["labels"] = np.array(gp.hyperpar_labels)
    out["alpha"] = gp.alpha
    out["mu"], out["sig"] = gp(pts)
    out["lml"] = np.array([gp.marginal_likelihood(t) for t in th])
    res = [gp.marginal_likelihood_gradient(t) for t in th]
    out["lml2"], out["grad"] = np.array([r[0] for r in res]), np.array([r[1] for r in res])
    res = [gp.loo_likelihood_gradient(t) for t in th]
    out["loo"], out["loo_grad"] = np.array([r[0] for r in res]), np.array([r[1] for r in res])
    return out


def case_plugin():
    """A covariance function that only implements the ABC (covariance.py:8-44) through the reference's classes."""
    from inference.gp import GpOptimiser
    from inference.gp.covariance import CovarianceFunction

    class Matern32(wl.Matern32Math, CovarianceFunction):
        pass

    out = {}
    x, y, e, pts, thetas = wl.plugin_problem()
    gp = GpRegressor(x, y, y_err=e, hyperpars=thetas[0], kernel=Matern32)
    out["thetas"] = thetas
    out["labels"] = np.array(gp.hyperpar_labels)
    out["bounds"] = np.array(gp.hp_bounds, dtype=float)
    out["K_xx"], out["L"], out["alpha"] = gp.K_xx, gp.L, gp.alpha
    out["mu"], out["sig"] = gp(pts)
    out["post_mu"], out["post_cov"] = gp.build_posterior(pts[:12])
    out["grad_mu"], out["grad_cov"] = gp.gradient(pts[:12])
    out["sd_mu"], out["sd_var"] = gp.spatial_derivatives(pts[:12])
    out["loo_mu"], out["loo_sig"] = gp.loo_predictions()
    out["lml"] = np.array([gp.marginal_likelihood(t) for t in thetas])
    res = [gp.marginal_likelihood_gradient(t) for t in thetas]
    out["lml2"], out["grad"] = np.array([r[0] for r in res]), np.array([r[1] for r in res])
    out["loo"] = np.array([gp.loo_likelihood(t) for t in thetas])
    res = [gp.loo_likelihood_gradient(t) for t in thetas]
    out["loo2"], out["loo_grad"] = np.array([r[0] for r in res]), np.array([r[1] for r in res])
    np.random.seed(17)
    gps = GpRegressor(x, y, y_err=e, kernel=Matern32, n_starts=3)
    out["search_theta"] = np.array(gps.hyperpars)
    out["search_lml"] = np.array(gps.marginal_likelihood(gps.hyperpars))
    # Bayesian optimisation on the plugin kernel: EI with its analytic gradient through gradient_terms
    bx, by, bounds = bo_problem()
    th = np.array([by.mean(), np.log(by.std()), np.log(2.0), np.log(2.0)])
    opt = GpOptimiser(bx, by, bounds=bounds, hyperpars=th, kernel=Matern32)
    np.random.seed(41)
    prop = opt.propose_evaluation()
    out["bo_theta"], out["bo_proposal"] = th, np.array(prop)
    out["bo_value"] = np.array(float(np.squeeze(opt.acquisition.opt_func(np.array(prop)))))
    return out


def case_means():
    """LinearMean / QuadraticMean (mean.py:54-126) with SE and RQ kernels on a 60-point 2-D set with a trend."""
    from inference.gp import LinearMean, QuadraticMean

    out = {}
    n, d = 60, 2
    rng = np.random.default_rng(606)
    x = rng.uniform(-1, 2, (n, d))
    y = 1.5 + 0.8 * x[:, 0] - 0.5 * x[:, 1] + 0.3 * x[:, 0] ** 2 + np.sin(3 * x[:, 0]) * np.cos(2 * x[:, 1]) + 0.05 * rng.normal(size=n)
    e = np.full(n, 0.05)
    pts = rng.uniform(-1, 2, (25, d))
    out.update(x=x, y=y, y_err=e, pts=pts)
    for mtag, mean, nm in (("lin", LinearMean, 1 + d), ("quad", QuadraticMean, 1 + 2 * d)):
        for ktag, kid in (("se", wl.SE), ("rq", wl.RQ)):
            key = f"{mtag}_{ktag}"
            trng = np.random.default_rng(77 + nm + kid)
            mean_part = np.concatenate([[y.mean()], 0.3 * trng.standard_normal(nm - 1)])
            cov_part = [np.log(y.std())] + ([0.5] if kid == wl.RQ else []) + [np.log(0.6)] * d
            base = np.concatenate([mean_part, cov_part])
            thetas = np.array([base + 0.1 * trng.standard_normal(base.size) for _ in range(3)])
            gp = GpRegressor(x, y, y_err=e, hyperpars=thetas[0], kernel=kernel_cls(kid), mean=mean)
            out[key + "_thetas"] = thetas
            out[key + "_labels"] = np.array(gp.hyperpar_labels)
            out[key + "_bounds"] = np.array(gp.hp_bounds, dtype=float)
            out[key + "_alpha"] = gp.alpha
            out[key + "_mu_train"] = gp.mu
            mu, sig = gp(pts)
            out[key + "_mu"], out[key + "_sig"] = mu, sig
            pm, pc = gp.build_posterior(pts[:10])
            out[key + "_post_mu"], out[key + "_post_cov"] = pm, pc
            out[key + "_lml"] = np.array([gp.marginal_likelihood(t) for t in thetas])
            res = [gp.marginal_likelihood_gradient(t) for t in thetas]
            out[key + "_lml2"] = np.array([r[0] for r in res])
            out[key + "_grad"] = np.array([r[1] for r in res])
            out[key + "_loo"] = np.array([gp.loo_likelihood(t) for t in thetas])
            res = [gp.loo_likelihood_gradient(t) for t in thetas]
            out[key + "_loo_grad"] = np.array([r[1] for r in res])
            if kid == wl.SE:
                sm, sv = gp.spatial_derivatives(pts[:10])
                out[key + "_sd_mu"], out[key + "_sd_var"] = sm, sv
    return out


def _oracle_pin_check():
    """The 16k cases come from the row-chunked oracle; before writing them, re-check the oracle against the imported
    reference on the same code paths (K bit-exact, factor-derived quantities to 1e-11) at a size the reference holds."""
    from oracle import gp_oracle as orc

    for cfg, kid, n, d in ((2, wl.SE, 2048, 8), (3, wl.RQ, 1536, 16)):
        x, y, e = wl.synthetic_dataset(cfg, n, d)
        th = wl.timing_theta(kid, y, d)
        ref = GpRegressor(x, y, y_err=e, hyperpars=th, kernel=kernel_cls(kid))
        o = orc.OracleGp(x, y, e, kernel=kid, hyperpars=th)
        assert np.array_equal(o.K_xx, ref.K_xx), "oracle K differs from the reference bit-wise"
        assert np.abs(o.alpha - ref.alpha).max() <= 1e-11 * np.abs(ref.alpha).max()
        assert abs(o.marginal_likelihood(th) - ref.marginal_likelihood(th)) <= 1e-11 * abs(ref.marginal_likelihood(th))


def _oracle_fit_outputs(o, th, pts, out, n_post=16):
    ii = idx64(o.n)
    out["theta"] = th
    out["alpha_norm"] = np.linalg.norm(o.alpha)
    out["alpha_idx"] = ii
    out["alpha_sub"] = o.alpha[ii]
    out["diagL_sub"] = np.diagonal(o.L)[ii]
    out["logdet"] = np.log(np.diagonal(o.L)).sum()
    r = o.y - th[0]
    out["lml_fit"] = -0.5 * float(r @ o.alpha) - out["logdet"]
    mu, sig = o(pts)
    out["pts"], out["mu"], out["sig"] = pts, mu, sig
    if n_post:
        pm, pc = o.build_posterior(pts[:n_post])
        out["post_mu"], out["post_cov"] = pm, pc


def case_head16k():
    """SE, N = 16384, d = 8 - the configuration BASELINE.json's metric is quoted on (regression.py:218-244,188-216,
    528-542), from the row-chunked oracle."""
    from oracle import gp_oracle as orc

    _oracle_pin_check()
    out = {}
    n, d = 16384, 8
    x, y, e = wl.synthetic_dataset(2, n, d)
    thetas = wl.theta_set(wl.SE, y, d, 3)
    pts = wl.query_points(2, 64, d)
    o = orc.OracleGp(x, y, e, kernel=orc.SE, hyperpars=thetas[0])
    _oracle_fit_outputs(o, thetas[0], pts, out)
    lml = [out["lml_fit"]]
    ld = [out["logdet"]]
    del o.K_xx
    for t in thetas[1:]:
        K = o._K(t[1:])
        Lf = np.linalg.cholesky(K)
        del K
        from scipy.linalg import solve_triangular

        v = solve_triangular(Lf, y - t[0], lower=True)
        ld.append(np.log(np.diagonal(Lf)).sum())
        lml.append(-0.5 * float(v @ v) - ld[-1])
        del Lf
    out["thetas"] = thetas
    out["lml"] = np.array(lml)
    out["logdets"] = np.array(ld)
    out["meta"] = np.array([2, wl.SE, n, d])
    return out


def case_cfg3_16k():
    """RQ, N = 16384, d = 16 (BASELINE config 3): the 8 grid points one GPU of eight evaluates, from the row-chunked
    oracle (regression.py:528-542; covariance.py:343-348)."""
    from oracle import gp_oracle as orc
    from scipy.linalg import solve_triangular

    _oracle_pin_check()
    out = {}
    n, d = 16384, 16
    x, y, e = wl.synthetic_dataset(3, n, d)
    grid = wl.theta_grid_cfg3(y, d)
    sel = np.arange(0, 64, 9)  # the grid's diagonal: every amplitude and every length scale once
    pts = wl.query_points(3, 64, d)
    o = orc.OracleGp(x, y, e, kernel=orc.RQ, hyperpars=grid[sel[0]])
    _oracle_fit_outputs(o, grid[sel[0]], pts, out, n_post=0)
    lml, ld = [out["lml_fit"]], [out["logdet"]]
    del o.K_xx, o.L
    for t in grid[sel[1:]]:
        K = o._K(t[1:])
        Lf = np.linalg.cholesky(K)
        del K
        v = solve_triangular(Lf, y - t[0], lower=True)
        ld.append(np.log(np.diagonal(Lf)).sum())
        lml.append(-0.5 * float(v @ v) - ld[-1])
        del Lf
    out["grid_idx"] = sel
    out["thetas"] = grid[sel]
    out["lml"] = np.array(lml)
    out["logdets"] = np.array(ld)
    out["meta"] = np.array([3, wl.RQ, n, d])
    return out


def case_gradpin():
    """The imported reference's LML and LOO gradients (regression.py:489-526,544-567) at sizes between the 32-point test
    set and config 2 - SE N = 2048 d = 8, RQ N = 1536 d = 16, at the second theta of theta_set - which pin the oracle's
    one-matrix-at-a-time gradients (tests/test_oracle_golden.py) before head16kg relies on them."""
    out = {}
    for tag, cfg, kid, n, d in (("se", 2, wl.SE, 2048, 8), ("rq", 3, wl.RQ, 1536, 16)):
        x, y, e = wl.synthetic_dataset(cfg, n, d)
        t = wl.theta_set(kid, y, d, 2)[1]
        gp = GpRegressor(x, y, y_err=e, hyperpars=t, kernel=kernel_cls(kid))
        out[tag + "_theta"] = t
        out[tag + "_lml_g_val"], out[tag + "_lml_g_grad"] = gp.marginal_likelihood_gradient(t)
        out[tag + "_loo_g_val"], out[tag + "_loo_g_grad"] = gp.loo_likelihood_gradient(t)
    return out


def case_cfg2g():
    """BASELINE config 2 (SE, N = 8192, d = 8): the imported reference's LML gradient at two thetas and its LOO value and
    gradient at one (regression.py:468-526,544-567) - the sizes above 4096 take a different chain of device kernels
    (one lane, k-skipped SYRK, L^-T by TRSM), and the round-4 timings of the gradient are quoted there."""
    out = {}
    x, y, e = wl.synthetic_dataset(2, 8192, 8)
    thetas = wl.theta_set(wl.SE, y, 8, 2)
    gp = GpRegressor(x, y, y_err=e, hyperpars=thetas[0], kernel=SquaredExponential)
    res = [gp.marginal_likelihood_gradient(t) for t in thetas]
    out["thetas"] = thetas
    out["lml_g_val"] = np.array([r[0] for r in res])
    out["lml_g_grad"] = np.array([r[1] for r in res])
    out["loo"] = np.array([gp.loo_likelihood(thetas[1])])
    v, g = gp.loo_likelihood_gradient(thetas[1])
    out["loo_g_val"], out["loo_g_grad"] = np.array([v]), np.array([g])
    lm, ls = gp.loo_predictions()
    ii = idx64(8192)
    out["loo_idx"], out["loo_mu_sub"], out["loo_sig_sub"] = ii, lm[ii], ls[ii]
    out["meta"] = np.array([2, wl.SE, 8192, 8])
    return out


def case_head16kg():
    """The metric's own size (SE, N = 16384, d = 8): LML gradient at two thetas, LOO value and gradient at one, from the
    oracle's one-matrix-at-a-time forms (the reference's list of d + 1 gradient matrices is 19 GB there, beside 34 GB of
    difference tensors), which tests/test_oracle_golden.py pins to the imported reference at N = 32 ... 2048 and which
    this script re-checks against the reference at N = 2048 / 1536 (case gradpin) before writing anything."""
    from oracle import gp_oracle as orc

    pin = case_gradpin()
    for tag, cfg, kid, n, d in (("se", 2, wl.SE, 2048, 8), ("rq", 3, wl.RQ, 1536, 16)):
        x, y, e = wl.synthetic_dataset(cfg, n, d)
        o = orc.OracleGp(x, y, e, kernel=kid)
        for nm, fn in (("lml", o.marginal_likelihood_gradient_lean), ("loo", o.loo_likelihood_gradient_lean)):
            v, g = fn(pin[tag + "_theta"])
            assert abs(v - pin[f"{tag}_{nm}_g_val"]) <= 1e-11 * abs(pin[f"{tag}_{nm}_g_val"])
            assert np.abs(g - pin[f"{tag}_{nm}_g_grad"]).max() <= 1e-10 * np.abs(pin[f"{tag}_{nm}_g_grad"]).max()
    out = {}
    n, d = 16384, 8
    x, y, e = wl.synthetic_dataset(2, n, d)
    thetas = wl.theta_set(wl.SE, y, d, 2)
    o = orc.OracleGp(x, y, e, kernel=orc.SE)
    res = [o.marginal_likelihood_gradient_lean(t) for t in thetas]
    out["thetas"] = thetas
    out["lml_g_val"] = np.array([r[0] for r in res])
    out["lml_g_grad"] = np.array([r[1] for r in res])
    v, g = o.loo_likelihood_gradient_lean(thetas[1])
    out["loo_g_val"], out["loo_g_grad"] = np.array([v]), np.array([g])
    out["meta"] = np.array([2, wl.SE, n, d])
    return out


IDX_TOMO = np.arange(0, 400, 7)  # 58 rows / columns of the 400 x 400 posterior covariance

CASES = {
    "search": case_search,
    "cpx": case_cpx,
    "cp4": case_cp4,
    "plugin": case_plugin,
    "means": case_means,
    "cp": case_cp,
    "het": case_het,
    "linv": case_linv,
    "linvp": case_linvp,
    "pt": case_pt,
    "t32": case_t32,
    "cfg1": case_cfg1,
    "rq256": case_rq256,
    "cfg4": case_cfg4,
    "cfg2": case_cfg2,
    "fail": case_fail,
    "head16k": case_head16k,
    "gradpin": case_gradpin,
    "cfg2g": case_cfg2g,
    "head16kg": case_head16kg,
    "cfg3_16k": case_cfg3_16k,
}

if __name__ == "__main__":
    names = sys.argv[1:] or ["t32", "cfg1", "rq256", "fail"]
    for nm in names:
        res = CASES[nm]()
        path = os.path.join(HERE, f"{nm}.npz")
        np.savez_compressed(path, **res)
        print(f"wrote {path}: {len(res)} arrays, {os.path.getsize(path)/1024:.1f} KiB")
