import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor, RationalQuadratic
def t(fn, reps=5):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3
mode = sys.argv[1]
head = None
x, y, e = wl.synthetic_dataset(2, 16384, 8)
head = GpRegressor(x, y, y_err=e, hyperpars=wl.timing_theta(wl.SE, y, 8))
if "c3" in mode:
    x3, y3, e3 = wl.synthetic_dataset(3, 8192, 16)
    g3 = GpRegressor(x3, y3, y_err=e3, hyperpars=wl.theta_grid_cfg3(y3, 16)[0], kernel=RationalQuadratic)
    g3.engine.set_streams(2)
    g3.marginal_likelihood_batch(wl.theta_grid_cfg3(y3, 16)[:4])
    if "keep" not in mode: g3.engine.close()
if "c5" in mode:
    x5, y5, e5 = wl.synthetic_dataset(5, 2048, 4)
    g5 = GpRegressor(x5, y5, y_err=e5, hyperpars=wl.timing_theta(wl.SE, y5, 4))
    g5.marginal_likelihood_batch(wl.timing_theta(wl.SE, y5, 4) + 0.01 * np.random.default_rng(0).standard_normal((64, 6)))
    g5.engine.close()
if "prof" in mode:
    head.engine.profile_enable(True); head.set_hyperparameters(head.hyperpars); head.engine.profile_enable(0)
if "closehead" in mode:
    head.engine.close()
x, y, e = wl.synthetic_dataset(2, 8192, 8)
th = wl.timing_theta(wl.SE, y, 8)
gp = GpRegressor(x, y, y_err=e, hyperpars=th)
gp.prepare_gradient()
print(mode, "fit %.2f" % t(lambda: gp.set_hyperparameters(th)), "lml %.2f" % t(lambda: gp.marginal_likelihood(th)),
      "grad %.2f" % t(lambda: gp.marginal_likelihood_gradient(th), 3), "lml again %.2f" % t(lambda: gp.marginal_likelihood(th)),
      "fit again %.2f" % t(lambda: gp.set_hyperparameters(th)))
