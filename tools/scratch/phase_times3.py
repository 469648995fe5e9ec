import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor
N, d, M = 16384, 8, 1024
x, y, e = wl.synthetic_dataset(2, N, d)
theta = wl.timing_theta(wl.SE, y, d)
pts = wl.query_points(2, M, d)
gp = GpRegressor(x, y, y_err=e, hyperpars=theta)
mode = sys.argv[1]
for rep in range(8):
    t0 = time.perf_counter(); gp.set_hyperparameters(theta); t1 = time.perf_counter()
    if mode == "fitpred":
        gp.engine.predict(pts)
    elif mode == "fitcall":
        gp(pts)
    elif mode == "fitlml":
        gp.marginal_likelihood(theta)
    t2 = time.perf_counter()
    print(f"[{mode}] fit {1e3*(t1-t0):.1f} ms | other {1e3*(t2-t1):.1f} ms")
    if mode == "norm":
        gp(pts)
        r = np.array([gp._logdet, float(np.linalg.norm(gp.alpha))])
        print(f"   [norm] total {1e3*(time.perf_counter()-t0):.1f}")
    if mode == "reset":
        gp(pts)
        if rep == 2:
            gp.engine.profile_reset(); gp.engine.sync()
        print(f"   [reset] total {1e3*(time.perf_counter()-t0):.1f}")
    if mode == "enable":
        gp(pts)
        if rep == 2:
            gp.engine.profile_enable(2 << 1); gp.engine.profile_reset(); gp.engine.sync()
        print(f"   [enable] total {1e3*(time.perf_counter()-t0):.1f}")
