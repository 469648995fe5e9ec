"""Host wall-time of the phases of one bench step (debug helper)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor
N, d, M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 8, 1024
x, y, e = wl.synthetic_dataset(2, N, d)
theta = wl.timing_theta(wl.SE, y, d)
pts = wl.query_points(2, M, d)
gp = GpRegressor(x, y, y_err=e, hyperpars=theta)
gp(pts)
for rep in range(10):
    t0 = time.perf_counter(); gp.set_hyperparameters(theta); t1 = time.perf_counter()
    mu, var = gp.engine.predict(pts); t2 = time.perf_counter()
    mu2, _ = gp.engine.predict(pts, want_var=False); t3 = time.perf_counter()
    lml = gp.marginal_likelihood(theta); t4 = time.perf_counter()
    print(f"fit {1e3*(t1-t0):.1f} ms | predict(mu,var) {1e3*(t2-t1):.1f} ms | predict(mu only) {1e3*(t3-t2):.1f} ms | lml {1e3*(t4-t3):.1f} ms")
import torch
torch.cuda.set_device(0)
for rep in range(10):
    t0 = time.perf_counter(); gp.set_hyperparameters(theta); t1 = time.perf_counter()
    mu, sig = gp(pts); t2 = time.perf_counter()
    res = torch.tensor([gp._logdet, float(np.linalg.norm(gp.alpha)), float(mu.sum()), float(sig.sum())], dtype=torch.float64, device="cuda")
    t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"[with torch] fit {1e3*(t1-t0):.1f} ms | gp(pts) {1e3*(t2-t1):.1f} ms | torch.tensor {1e3*(t3-t2):.1f} ms | sync {1e3*(t4-t3):.1f}")
