import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
x, y, e = wl.synthetic_dataset(2, N, 8)
th = wl.timing_theta(wl.SE, y, 8)
gp = GpRegressor(x, y, y_err=e, hyperpars=th)
if os.environ.get("GPMI_NO_PREPARE") is None:
    t0 = time.perf_counter(); gp.prepare_gradient(); print(f"prepare_gradient {1e3*(time.perf_counter()-t0):.1f} ms (workspaces of the gradient path, once)")
for rep in range(3):
    t0 = time.perf_counter(); v, g = gp.marginal_likelihood_gradient(th); t1 = time.perf_counter()
    l = gp.marginal_likelihood(th); t2 = time.perf_counter()
    print(f"N={N}: LML+grad {1e3*(t1-t0):.1f} ms ({N**3/(t1-t0)/1e12:.1f} TFLOP/s of N^3) | LML {1e3*(t2-t1):.1f} ms")
