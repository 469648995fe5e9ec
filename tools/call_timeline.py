"""Kernel timeline of ONE steady-state call of a public method at N (under rocprofv3 --kernel-trace; tools/trace_last_call.py prints
the last call): usage: python tools/call_timeline.py <fit|lml|grad|predict> [N] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor
what = sys.argv[1] if len(sys.argv) > 1 else "grad"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
x, y, e = wl.synthetic_dataset(2, n, 8)
th = wl.timing_theta(wl.SE, y, 8)
pts = wl.query_points(2, 1024, 8)
gp = GpRegressor(x, y, y_err=e, hyperpars=th)
gp.prepare_gradient()
fn = {"fit": lambda: gp.set_hyperparameters(th), "lml": lambda: gp.marginal_likelihood(th),
      "grad": lambda: gp.marginal_likelihood_gradient(th), "predict": lambda: gp(pts)}[what]
for _ in range(5):
    fn()
t0 = time.perf_counter()
for _ in range(reps):
    fn()
print(f"{what} N={n}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms")
time.sleep(0.05)
fn()  # the call the timeline tool prints: the kernels behind the last 50 ms pause
gp.engine.close()
