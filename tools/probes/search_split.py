"""Host vs device in the constructor's hyper-parameter search (N = 2048, d = 4, 5 L-BFGS-B starts in lockstep; LML and
cross-validation objective): cProfile of one construction."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor
n, d = 2048, 4
x, y, e = wl.synthetic_dataset(5, n, d)
for cv in (False, True):
    np.random.seed(1); GpRegressor(x, y, y_err=e, cross_val=cv)
    np.random.seed(1); t0 = time.perf_counter(); GpRegressor(x, y, y_err=e, cross_val=cv); dt = time.perf_counter() - t0
    print(f"cross_val={cv}: constructor {dt*1e3:.1f} ms")
    pr = cProfile.Profile(); np.random.seed(1); pr.enable(); GpRegressor(x, y, y_err=e, cross_val=cv); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(8)
