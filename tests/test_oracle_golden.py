"""
Pins the CPU oracle (oracle/gp_oracle.py) against golden vectors produced by
the imported reference (tests/golden/make_golden.py).  CPU only.

Tolerances: the oracle uses the same LAPACK entry points as the reference, so
agreement is at the oracle noise floor (different BLAS blocking / thread counts
between the two call sequences): rtol 1e-11 scaled by the quantity's magnitude.
"""
import numpy as np
import pytest

from oracle import gp_oracle as orc
import workloads as wl


def close(a, b, rtol=1e-11, scale=None):
    a, b = np.asarray(a, float), np.asarray(b, float)
    s = np.abs(b).max() if scale is None else scale
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.abs(a - b).max() <= rtol * max(s, 1e-300), np.abs(a - b).max() / max(s, 1e-300)


@pytest.mark.parametrize("name,kid", [("se_", orc.SE), ("rq_", orc.RQ)])
def test_t32_all_methods(golden, name, kid):
    g = golden("t32")
    x, y, e = g["x"], g["y"], g["y_err"]
    th = g[name + "thetas"]
    gp = orc.OracleGp(x, y, e, kernel=kid, hyperpars=th[0])
    # bit-exact K (same ops, same order) — covariance.py:247-255 / 343-348
    assert np.array_equal(gp.K_xx, g[name + "K_xx"])
    close(gp.L, g[name + "L"])
    close(gp.alpha, g[name + "alpha"])
    close(np.array(gp.hp_bounds), g[name + "hp_bounds"], rtol=1e-13)
    close([gp.marginal_likelihood(t) for t in th], g[name + "lml"])
    res = [gp.marginal_likelihood_gradient(t) for t in th]
    close([r[0] for r in res], g[name + "lml_g_val"])
    close([r[1] for r in res], g[name + "lml_g_grad"], rtol=1e-10)
    mu, sig = gp(g[name + "pts"])
    close(mu, g[name + "mu"])
    close(sig, g[name + "sig"])
    pm, pc = gp.build_posterior(g[name + "pts"][:16])
    close(pm, g[name + "post_mu"])
    close(pc, g[name + "post_cov"])
    K, dK = orc.kernel_build_and_grads(kid, x, th[1][1:])
    assert np.array_equal(K, g[name + "cov_K"])
    close(np.array(dK), g[name + "cov_dK"], rtol=1e-14)
    assert np.array_equal(orc.kernel_cross(kid, g[name + "pts"], x, th[1][1:]), g[name + "cov_cross"])
    lm, ls = gp.loo_predictions()
    close(lm, g[name + "loo_mu"])
    close(ls, g[name + "loo_sig"])
    close([gp.loo_likelihood(t) for t in th], g[name + "loo"])
    res = [gp.loo_likelihood_gradient(t) for t in th]
    close([r[0] for r in res], g[name + "loo_g_val"])
    close([r[1] for r in res], g[name + "loo_g_grad"], rtol=1e-10)


@pytest.mark.parametrize("name,kid", [("se_", orc.SE), ("rq_", orc.RQ)])
def test_lean_gradients_equal_the_reference(golden, name, kid):
    """The one-matrix-at-a-time gradients (the forms make_golden.py runs at N = 16384, where the reference's list of
    gradient matrices does not fit) against the imported reference's regression.py:489-526,544-567."""
    g = golden("t32")
    th = g[name + "thetas"]
    gp = orc.OracleGp(g["x"], g["y"], g["y_err"], kernel=kid, hyperpars=th[0])
    res = [gp.marginal_likelihood_gradient_lean(t) for t in th]
    close([r[0] for r in res], g[name + "lml_g_val"])
    close([r[1] for r in res], g[name + "lml_g_grad"], rtol=1e-10)
    res = [gp.loo_likelihood_gradient_lean(t) for t in th]
    close([r[0] for r in res], g[name + "loo_g_val"])
    close([r[1] for r in res], g[name + "loo_g_grad"], rtol=1e-10)
    # and bit for bit the oracle's list-of-matrices form of the LML gradient (same expressions, same sums)
    for t in th[:2]:
        a, b = gp.marginal_likelihood_gradient(t), gp.marginal_likelihood_gradient_lean(t)
        assert a[0] == b[0] and np.array_equal(a[1], b[1])


def test_lean_gradients_white_noise_and_mid_size(golden):
    g = golden("t32")
    th = g["sewn_theta"]
    gp = orc.OracleGp(g["x"], g["y"], g["y_err"], kernel=orc.SE, hyperpars=th, white_noise=True)
    v, gr = gp.marginal_likelihood_gradient_lean(th)
    close(v, g["sewn_lml_g_val"])
    close(gr, g["sewn_lml_g_grad"], rtol=1e-10)
    # the reference at N = 1536 (RQ, d = 16) and N = 2048 (SE, d = 8): tests/golden/make_golden.py gradpin
    g = golden("gradpin")
    for tag, cfg, kid, n, d in (("se", 2, orc.SE, 2048, 8), ("rq", 3, orc.RQ, 1536, 16)):
        x, y, e = wl.synthetic_dataset(cfg, n, d)
        t = g[tag + "_theta"]
        gp = orc.OracleGp(x, y, e, kernel=kid)
        v, gr = gp.marginal_likelihood_gradient_lean(t)
        close(v, g[tag + "_lml_g_val"])
        close(gr, g[tag + "_lml_g_grad"], rtol=1e-10)
        v, gr = gp.loo_likelihood_gradient_lean(t)
        close(v, g[tag + "_loo_g_val"])
        close(gr, g[tag + "_loo_g_grad"], rtol=1e-10)


def test_t32_spatial_gradients_and_acquisition(golden):
    g = golden("t32")
    gp = orc.OracleGp(g["x"], g["y"], g["y_err"], kernel=orc.SE, hyperpars=g["se_thetas"][0])
    pts = g["se_pts"]
    gm, gc = gp.gradient(pts[:16])
    close(gm, g["se_grad_mu"])
    close(gc, g["se_grad_cov"])
    sm, sv = gp.spatial_derivatives(pts[:16])
    close(sm, g["se_sd_mu"])
    close(sv, g["se_sd_var"])
    mu, sig = gp(pts)
    mu_max = g["y"].max()
    close(orc.ei_value(mu, sig, mu_max), g["se_ei_call"])
    close(-orc.ei_log(mu, sig, mu_max), g["se_ei_opt"])
    sm, sv = gp.spatial_derivatives(pts)
    val, grad = orc.ei_opt_func_gradient(mu, sig, sm, sv, mu_max)
    close(val, g["se_ei_optg_val"])
    close(grad, g["se_ei_optg_grad"], rtol=1e-10)
    # UCB (acquisition.py:168-189, kappa = 2) and MaxVariance (acquisition.py:212-229)
    close(mu + 2.0 * sig, g["se_ucb_call"])
    close(-(dmu := sm) - 0.5 * 2.0 * sv / sig[:, None], g["se_ucb_optg_grad"], rtol=1e-10)
    close(sig**2, g["se_mv_call"])
    close(-sv, g["se_mv_optg_grad"], rtol=1e-10)


def test_t32_white_noise_composite(golden):
    g = golden("t32")
    th = g["sewn_theta"]
    gp = orc.OracleGp(g["x"], g["y"], g["y_err"], kernel=orc.SE, hyperpars=th, white_noise=True)
    close(np.array(gp.hp_bounds), g["sewn_hp_bounds"], rtol=1e-13)
    close(gp.marginal_likelihood(th), g["sewn_lml"])
    v, gr = gp.marginal_likelihood_gradient(th)
    close(v, g["sewn_lml_g_val"])
    close(gr, g["sewn_lml_g_grad"], rtol=1e-10)
    close(gp.alpha, g["sewn_alpha"])
    mu, sig = gp(g["se_pts"])
    close(mu, g["sewn_mu"])
    close(sig, g["sewn_sig"])


def test_t32_one_dimensional(golden):
    g = golden("t32")
    gp = orc.OracleGp(g["d1_x"], g["d1_y"], g["d1_err"], kernel=orc.SE, hyperpars=g["d1_theta"])
    close(np.array(gp.hp_bounds), g["d1_hp_bounds"], rtol=1e-13)
    mu, sig = gp(g["d1_pts"])
    close(mu, g["d1_mu"])
    close(sig, g["d1_sig"])
    gm, gc = gp.gradient(g["d1_pts"])
    close(gm, g["d1_grad_mu"])
    close(gc, g["d1_grad_cov"])
    sm, sv = gp.spatial_derivatives(g["d1_pts"])
    close(sm, g["d1_sd_mu"])
    close(sv, g["d1_sd_var"])


@pytest.mark.parametrize("case", ["cfg1", "rq256", "cfg4", "cfg2"])
def test_synthetic_configs(golden, case):
    g = golden(case)
    cfg, kid, n, d = [int(v) for v in g["meta"]]
    if n > 4096:
        pytest.skip("cfg2-size oracle run is exercised by the GPU parity test and bench, not the CPU suite")
    x, y, e = wl.synthetic_dataset(cfg, n, d)
    th = g["thetas"]
    assert np.array_equal(th, wl.theta_set(kid, y, d, len(th)))
    gp = orc.OracleGp(x, y, e, kernel=kid, hyperpars=th[0])
    close(np.array(gp.hp_bounds), g["hp_bounds"], rtol=1e-12)
    n_lml = len(th) if n <= 1024 else 1
    close([gp.marginal_likelihood(t) for t in th[:n_lml]], g["lml"][:n_lml])
    if "lml_g_val" in g and n <= 1024:
        res = [gp.marginal_likelihood_gradient(t) for t in th[:2]]
        close([r[0] for r in res], g["lml_g_val"][:2])
        close([r[1] for r in res], g["lml_g_grad"][:2], rtol=1e-9)
    close(np.linalg.norm(gp.alpha), g["alpha_norm"])
    close(gp.alpha[g["alpha_idx"]], g["alpha_sub"], rtol=1e-10)
    close(np.diagonal(gp.L)[g["alpha_idx"]], g["diagL_sub"])
    close(np.log(np.diagonal(gp.L)).sum(), g["logdet"])
    mu, sig = gp(g["pts"])
    close(mu, g["mu"], rtol=1e-10)
    close(sig, g["sig"], rtol=1e-10)
    pm, pc = gp.build_posterior(g["pts"][:16])
    close(pm, g["post_mu"], rtol=1e-10)
    close(pc, g["post_cov"], rtol=1e-10)
    if "ei_call" in g:
        m2, s2 = gp(g["cand"])
        close(orc.ei_value(m2, s2, float(g["mu_max"])), g["ei_call"], rtol=1e-9)
        close(-orc.ei_log(m2, s2, float(g["mu_max"])), g["ei_opt"], rtol=1e-9)
        far = float(g["ei_far_mu_max"])
        close(-orc.ei_log(m2[:200], s2[:200], far), g["ei_far_opt"], rtol=1e-9)
        sm, sv = gp.spatial_derivatives(g["cand"][:200])
        val, grad = orc.ei_opt_func_gradient(m2[:200], s2[:200], sm, sv, far)
        close(val, g["ei_far_optg_val"], rtol=1e-9)
        close(grad, g["ei_far_optg_grad"], rtol=1e-8)


def test_cholesky_failure_sentinel(golden):
    g = golden("fail")
    gp = orc.OracleGp(g["x"], g["y"], y_cov=g["y_cov"], kernel=orc.SE)
    assert gp.marginal_likelihood(g["theta_bad"]) == -1e50 == float(g["lml_bad"])
    close(gp.marginal_likelihood(g["theta_ok"]), g["lml_ok"])


# ---------------------------------------------------------------------------------------
# GpLinearInverter (inversion.py): the oracle restatement against the reference's outputs
# ---------------------------------------------------------------------------------------
LINV_CASES = [(prob, tag, kid, wn) for prob in ("deconv", "tomo")
              for tag, kid, wn in (("se", orc.SE, False), ("rq", orc.RQ, False), ("sewn", orc.SE, True))]


@pytest.mark.parametrize("prob,tag,kid,wn", LINV_CASES)
def test_linear_inverter_oracle_matches_reference(golden, prob, tag, kid, wn):
    from oracle.linv_oracle import OracleLinearInverter

    g = golden("linv")
    pos, A, y, y_err = wl.linv_problem(prob)
    inv = OracleLinearInverter(y, y_err, A, pos, kid, white_noise=wn)
    idx = np.arange(0, 400, 7)
    for i, th in enumerate(wl.linv_thetas(prob, kid, wn)):
        key = f"{prob}_{tag}_{i}"
        assert np.array_equal(th, g[key + "_theta"])
        close(inv.marginal_likelihood(th), g[key + "_lml"], rtol=1e-11)
        lml, grad = inv.marginal_likelihood_gradient(th)
        close(lml, g[key + "_lml2"], rtol=1e-11)
        close(grad, g[key + "_grad"], rtol=1e-9)
        pm, pc = inv.calculate_posterior(th)
        close(pm, g[key + "_pmean"], rtol=1e-9)
        close(inv.calculate_posterior_mean(th), g[key + "_pmean_only"], rtol=1e-9)
        close(pc if prob == "deconv" else pc[idx][:, idx], g[key + "_pcov"], rtol=1e-9)
