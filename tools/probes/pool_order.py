"""(Record of the experiment in profiles/HISTORY.md R6.18: the GPMI_PAIR_MIN_TILES / GPMI_PAIR_POOL_SMALL switches and the pool it
exercised were a temporary patch of api.hip and are not in the library.)
Does a fit at N = 8192 depend on which pooled pair its handle got / on how many other handles (with pairs) are alive?
usage: python tools/probes/pool_order.py <alive small handles before the big one> [closed small handles before]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor
alive_n = int(sys.argv[1]); closed_n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
xs, ys, es = wl.synthetic_dataset(4, 2048, 4); ths = wl.timing_theta(wl.SE, ys, 4)
for _ in range(closed_n):
    g = GpRegressor(xs, ys, y_err=es, hyperpars=ths); g.engine.close()
alive = [GpRegressor(xs, ys, y_err=es, hyperpars=ths) for _ in range(alive_n)]
x, y, e = wl.synthetic_dataset(2, 8192, 8); th = wl.timing_theta(wl.SE, y, 8)
big = GpRegressor(x, y, y_err=e, hyperpars=th)
for _ in range(5): big.set_hyperparameters(th)
t0 = time.perf_counter()
for _ in range(40): big.set_hyperparameters(th)
print(f"{closed_n} small handles opened and closed, {alive_n} alive, then N=8192: fit {(time.perf_counter()-t0)/40*1e3:.3f} ms")
