"""Timings of the BASELINE.json configurations other than the headline (single GPU), through the public classes.
usage: python tools/config_bench.py [cfg2] [cfg3] [cfg4] [cfg5]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor, RationalQuadratic, ExpectedImprovement

which = sys.argv[1:] or ["cfg2", "cfg3", "cfg4", "cfg5"]

def t(fn, reps=3, steady=0.1):
    """mean wall time of a call in steady state (bench.py: _timeit - the shader clock needs tens of milliseconds of load)"""
    t0 = time.perf_counter()
    fn()
    while time.perf_counter() - t0 < steady:
        fn()
    t0 = time.perf_counter()
    n = 0
    while n < reps or (time.perf_counter() - t0 < steady and n < 500):
        out = fn()
        n += 1
    return (time.perf_counter() - t0) / n, out

if "cfg2" in which:
    x, y, e = wl.synthetic_dataset(2, 8192, 8)
    th = wl.timing_theta(wl.SE, y, 8)
    pts = wl.query_points(2, 1024, 8)
    gp = GpRegressor(x, y, y_err=e, hyperpars=th)
    dt_fit, _ = t(lambda: gp.set_hyperparameters(th))
    dt_pred, _ = t(lambda: gp(pts))
    dt_lml, _ = t(lambda: gp.marginal_likelihood(th))
    dt_g, _ = t(lambda: gp.marginal_likelihood_gradient(th), reps=2)
    fl = 8192**3 / 3
    print(f"cfg2 SE N=8192 d=8: fit {dt_fit*1e3:.1f} ms ({fl/dt_fit/1e12:.1f} TFLOP/s potrf-equivalent) | predict M=1024 {dt_pred*1e3:.1f} ms | "
          f"LML {dt_lml*1e3:.1f} ms | LML+grad {dt_g*1e3:.1f} ms")
if "cfg3" in which:
    x, y, e = wl.synthetic_dataset(3, 16384, 16)
    grid = wl.theta_grid_cfg3(y, 16)
    gp = GpRegressor(x, y, y_err=e, hyperpars=grid[0], kernel=RationalQuadratic)
    gp.engine.set_streams(2)
    dt, vals = t(lambda: gp.marginal_likelihood_batch(grid[:8]), reps=1)
    print(f"cfg3 RQ N=16384 d=16: 8 LML evaluations (one GPU's share of the 64-point grid) {dt*1e3:.1f} ms = {dt/8*1e3:.1f} ms each, "
          f"{8*16384**3/3/dt/1e12:.1f} TFLOP/s; 64-grid on 1 GPU ~ {dt*8:.2f} s")
if "cfg4" in which:
    x, y, e = wl.synthetic_dataset(4, 4096, 4)
    th = wl.timing_theta(wl.SE, y, 4)
    cand = wl.query_points(4004, 1000, 4)
    gp = GpRegressor(x, y, y_err=e, hyperpars=th)
    ei = ExpectedImprovement(); ei.update_gp(gp)
    dt_v, _ = t(lambda: ei.call_batch(cand))
    dt_g, _ = t(lambda: ei.opt_func_gradient_batch(cand))
    print(f"cfg4 SE N=4096 d=4: EI at 1000 candidates {dt_v*1e3:.2f} ms | -ln EI + gradient at 1000 candidates {dt_g*1e3:.2f} ms")
if "cfg5" in which:
    x, y, e = wl.synthetic_dataset(5, 2048, 4)
    th = wl.timing_theta(wl.SE, y, 4)
    gp = GpRegressor(x, y, y_err=e, hyperpars=th)
    rng = np.random.default_rng(0)
    thetas = th + 0.05 * rng.standard_normal((512, th.size))
    for S in (1, 4, 16, 64):
        gp.engine.set_streams(S)
        dt, _ = t(lambda: gp.marginal_likelihood_batch(thetas), reps=2)
        print(f"cfg5 SE N=2048 d=4: 512 LML evaluations on {S:2d} streams: {dt*1e3:.1f} ms = {512/dt:.0f} evals/s "
              f"({512*2048**3/3/dt/1e12:.2f} TFLOP/s)")
if "linv" in which:
    # GpLinearInverter: m data values, n model parameters on a 2-D grid (synthetic smooth chords)
    from inference_amd.gp import GpLinearInverter
    m, g = 4096, 90
    n = g * g
    ax = (np.arange(g) + 0.5) / g
    X, Y = np.meshgrid(ax, ax, indexing="ij")
    pos = np.stack([X.ravel(), Y.ravel()], axis=1)
    rng = np.random.default_rng(7)
    ang, off = rng.uniform(0, np.pi, m), rng.uniform(-0.35, 0.35, m)
    dist = (pos[None, :, 0] - 0.5) * np.cos(ang)[:, None] + (pos[None, :, 1] - 0.5) * np.sin(ang)[:, None] - off[:, None]
    A = np.exp(-0.5 * (dist / 0.02) ** 2); A /= A.sum(axis=1, keepdims=True)
    truth = np.exp(-((pos[:, 0] - 0.4) ** 2 + (pos[:, 1] - 0.55) ** 2) / 0.03)
    y_err = np.full(m, 0.01); y = A @ truth + rng.normal(size=m) * y_err
    gli = GpLinearInverter(y=y, y_err=y_err, model_matrix=A, parameter_spatial_positions=pos)
    th = np.array([0.1, 0.0, np.log(0.2), np.log(0.2)])
    dt_l, _ = t(lambda: gli.marginal_likelihood(th))
    dt_g, _ = t(lambda: gli.marginal_likelihood_gradient(th), reps=2)
    dt_m, _ = t(lambda: gli.calculate_posterior_mean(th), reps=2)
    dt_p, _ = t(lambda: gli.calculate_posterior(th), reps=2)
    fl = 2.0 * m * n * n + 2.0 * m * m * n / 2 + m**3 / 3
    print(f"linv SE m={m} n={n} d=2: LML {dt_l*1e3:.1f} ms ({fl/dt_l/1e12:.1f} TFLOP/s) | LML+grad {dt_g*1e3:.1f} ms | "
          f"posterior mean {dt_m*1e3:.1f} ms | mean+covariance ({n}x{n} download) {dt_p*1e3:.1f} ms")
    if "--cpu" in sys.argv:
        import time as _t
        from oracle.linv_oracle import OracleLinearInverter
        sub = slice(0, 1024); nsub = 45 * 45
        o = OracleLinearInverter(y[sub], y_err[sub], A[sub, :nsub], pos[:nsub], wl.SE)
        t0 = _t.perf_counter(); o.marginal_likelihood(th); d1 = _t.perf_counter() - t0
        print(f"     CPU oracle at m=1024 n={nsub}: LML {d1*1e3:.0f} ms")
