import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor, RationalQuadratic
x, y, e = wl.synthetic_dataset(3, 16384, 16)
grid = wl.theta_grid_cfg3(y, 16)
gp = GpRegressor(x, y, y_err=e, hyperpars=grid[0], kernel=RationalQuadratic)
for S in (1, 2, 3, 4):
    gp.engine.set_streams(S)
    gp.marginal_likelihood_batch(grid[:S])
    t0 = time.perf_counter(); v = gp.marginal_likelihood_batch(grid[:12]); dt = time.perf_counter() - t0
    print(f"S={S}: 12 LML in {dt*1e3:.1f} ms = {dt/12*1e3:.2f} ms each ({12*16384**3/3/dt/1e12:.1f} TFLOP/s)")
