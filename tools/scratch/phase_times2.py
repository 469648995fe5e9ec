"""Same as phase_times.py but with torch's HIP runtime initialised first (as bench.py does)."""
import os, sys, time
import torch
torch.cuda.set_device(0)
_ = torch.zeros(1, device="cuda")
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
gp(pts)
for rep in range(3):
    t0 = time.perf_counter(); gp.set_hyperparameters(theta); t1 = time.perf_counter()
    mu, sig = gp(pts); t2 = time.perf_counter()
    res = torch.tensor([gp._logdet, float(np.linalg.norm(gp.alpha)), float(mu.sum()), float(sig.sum())], dtype=torch.float64, device="cuda")
    t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"[torch first] fit {1e3*(t1-t0):.1f} ms | gp(pts) {1e3*(t2-t1):.1f} ms | torch.tensor {1e3*(t3-t2):.1f} ms | sync {1e3*(t4-t3):.1f}")
os.system("grep -E 'libamdhip64|libhsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
