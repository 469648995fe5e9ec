"""Other public operations at N = 4096, d = 4 (posterior of 512 points, mean / covariance gradients at 200 points, LOO
likelihood + gradient, spatial derivatives at 1000 points): wall time each; run under rocprofv3 + tools/kstats.py for the kernels.
usage: python tools/probes/ops_profile.py [op ...]   (ops: posterior gradient loo sd lmlgrad)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor
n, d = 4096, 4
x, y, e = wl.synthetic_dataset(4, n, d)
th = wl.timing_theta(wl.SE, y, d)
gp = GpRegressor(x, y, y_err=e, hyperpars=th)
ops = sys.argv[1:] or ["posterior", "gradient", "loo", "sd", "lmlgrad"]
def t(fn, reps=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3
if "posterior" in ops: print(f"build_posterior(512 points): {t(lambda: gp.build_posterior(wl.query_points(4, 512, d))):.2f} ms")
if "gradient" in ops: print(f"gradient(200 points): {t(lambda: gp.gradient(wl.query_points(4, 200, d))):.2f} ms")
if "loo" in ops: print(f"loo_likelihood_gradient: {t(lambda: gp.loo_likelihood_gradient(th)):.2f} ms")
if "sd" in ops: print(f"spatial_derivatives(1000 points): {t(lambda: gp.spatial_derivatives(wl.query_points(4, 1000, d))):.2f} ms")
if "lmlgrad" in ops: print(f"marginal_likelihood_gradient: {t(lambda: gp.marginal_likelihood_gradient(th)):.2f} ms")
