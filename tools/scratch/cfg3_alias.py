"""Does a second live handle (the headline's regressor) slow the two-lane LML sweep of config 3?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor, RationalQuadratic
mode = sys.argv[1] if len(sys.argv) > 1 else "keep"
x8, y8, e8 = wl.synthetic_dataset(1, 16384, 8)
gp = GpRegressor(x8, y8, y_err=e8, hyperpars=wl.timing_theta(wl.SE, y8, 8))
gp(wl.query_points(1, 1024, 8))
if mode == "close":
    gp.engine.close()
x, y, e = wl.synthetic_dataset(3, 16384, 16)
grid = wl.theta_grid_cfg3(y, 16)
gp3 = GpRegressor(x, y, y_err=e, hyperpars=grid[0], kernel=RationalQuadratic)
gp3.engine.set_streams(2)
gp3.marginal_likelihood_batch(grid[:2])
for rep in range(2):
    t0 = time.perf_counter()
    v = gp3.marginal_likelihood_batch(grid[:16])
    dt = time.perf_counter() - t0
    print(f"{mode}: 16 evaluations {dt*1e3:.1f} ms = {dt/16*1e3:.2f} ms each")
