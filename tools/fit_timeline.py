"""Kernel timeline of ONE steady-state GpRegressor.set_hyperparameters (fit) at N (default 8192): run under
rocprofv3 --kernel-trace, then tools/timeline_full.py on the trace prints the last fit.  usage: python tools/fit_timeline.py [N] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
x, y, e = wl.synthetic_dataset(2, n, 8)
th = wl.timing_theta(wl.SE, y, 8)
gp = GpRegressor(x, y, y_err=e, hyperpars=th)
for _ in range(5):
    gp.set_hyperparameters(th)
t0 = time.perf_counter()
for _ in range(reps):
    gp.set_hyperparameters(th)
print(f"N={n}: fit {(time.perf_counter() - t0) / reps * 1e3:.3f} ms")
gp.engine.close()
