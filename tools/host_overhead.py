"""Host-side cost of the two public calls of a headline step (set_hyperparameters, __call__) on a problem small enough
that the device work is negligible, with the Python profile of where it goes.  usage: python tools/host_overhead.py [N] [M]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = int(sys.argv[2]) if len(sys.argv) > 2 else 128
x, y, e = wl.synthetic_dataset(1, n, 8)
th = wl.timing_theta(wl.SE, y, 8)
pts = wl.query_points(1, m, 8)
gp = GpRegressor(x, y, y_err=e, hyperpars=th)
for _ in range(20):
    gp.set_hyperparameters(th); gp(pts)
reps = 300
t0 = time.perf_counter()
for _ in range(reps):
    gp.set_hyperparameters(th)
t1 = time.perf_counter()
for _ in range(reps):
    gp(pts)
t2 = time.perf_counter()
print(f"N={n} M={m}: set_hyperparameters {1e6*(t1-t0)/reps:.0f} us per call, __call__ {1e6*(t2-t1)/reps:.0f} us per call")
pr = cProfile.Profile()
pr.enable()
for _ in range(reps):
    gp.set_hyperparameters(th); gp(pts)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
