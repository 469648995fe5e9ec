"""What a handle costs to open and close with and without the CU-masked stream pair (np >= 5120 has one), and the fit at
N = 4096 / 5120 (stream-ordered below 5120, look-ahead + flag-ordered tail from there)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor
for n in (4096, 4992, 5120, 6144):
    x, y, e = wl.synthetic_dataset(4, n, 4)
    th = wl.timing_theta(wl.SE, y, 4)
    t0 = time.perf_counter(); gp = GpRegressor(x, y, y_err=e, hyperpars=th); t1 = time.perf_counter()
    for _ in range(5): gp.set_hyperparameters(th)
    t2 = time.perf_counter()
    for _ in range(20): gp.set_hyperparameters(th)
    t3 = time.perf_counter()
    gp.engine.close(); t4 = time.perf_counter()
    print(f"N={n}: construct {1e3*(t1-t0):.1f} ms, fit {(t3-t2)/20*1e3:.3f} ms, close {1e3*(t4-t3):.1f} ms")
for n in (5120,):
    x, y, e = wl.synthetic_dataset(4, n, 4); th = wl.timing_theta(wl.SE, y, 4)
    t0 = time.perf_counter()
    for _ in range(3):
        gp = GpRegressor(x, y, y_err=e, hyperpars=th); gp.engine.close()
    print(f"N={n}: construct + close, three in a row: {(time.perf_counter()-t0)/3*1e3:.1f} ms each")
