"""cfg3: LML grid at N = 16384 d = 16 (RationalQuadratic) with 1 / 2 / 4 evaluation lanes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor, RationalQuadratic
x, y, e = wl.synthetic_dataset(3, 16384, 16)
grid = wl.theta_grid_cfg3(y, 16)
gp = GpRegressor(x, y, y_err=e, hyperpars=grid[0], kernel=RationalQuadratic)
ref = None
for S in (1, 2, 4, 1, 2):
    gp.engine.set_streams(S)
    gp.marginal_likelihood_batch(grid[:2])
    t0 = time.perf_counter()
    v = gp.marginal_likelihood_batch(grid[:16])
    dt = time.perf_counter() - t0
    if ref is None:
        ref = v
    print(f"streams {S}: 16 evaluations {dt*1e3:.1f} ms = {dt/16*1e3:.2f} ms each, {16*16384**3/3/dt/1e12:.1f} TFLOP/s, max |diff| vs first {np.abs(v-ref).max():.2e}")
