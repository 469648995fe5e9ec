"""Rate of the lockstep likelihood batch: T evaluations (default 512) at N = 2048, d = 4 (config 5's unit of work).
usage: python tools/lml_batch_time.py [T] [N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor
T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
x, y, e = wl.synthetic_dataset(5, N, 4)
th = wl.timing_theta(wl.SE, y, 4)
gp = GpRegressor(x, y, y_err=e, hyperpars=th)
thetas = th + 0.05 * np.random.default_rng(0).standard_normal((T, th.size))
gp.marginal_likelihood_batch(thetas)
t0 = time.perf_counter()
for _ in range(3): v = gp.marginal_likelihood_batch(thetas)
dt = (time.perf_counter() - t0) / 3
print(f"{T} LML evaluations at N={N}: {dt*1e3:.1f} ms = {T/dt:.0f} evals/s ({T/dt*N**3/3/1e12:.1f} TFLOP/s) checksum {v.sum():.10g}")
