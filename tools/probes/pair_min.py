"""(Record of the experiment in profiles/HISTORY.md R6.18: the GPMI_PAIR_MIN_TILES / GPMI_PAIR_POOL_SMALL switches and the pool it
exercised were a temporary patch of api.hip and are not in the library.)
Fit and LML at mid sizes with and without the CU-masked pair (GPMI_PAIR_MIN_TILES: handles of at least that many tile
rows get one - 40 by default, i.e. N >= 5120): what the flag-ordered tail would buy below that."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor
for n in (1536, 2048, 3072, 4096, 4992):
    x, y, e = wl.synthetic_dataset(4, n, 4)
    th = wl.timing_theta(wl.SE, y, 4)
    gp = GpRegressor(x, y, y_err=e, hyperpars=th)
    for _ in range(5): gp.set_hyperparameters(th)
    t2 = time.perf_counter()
    for _ in range(30): gp.set_hyperparameters(th)
    t3 = time.perf_counter()
    for _ in range(3): gp.marginal_likelihood(th)
    t4 = time.perf_counter()
    for _ in range(20): v = gp.marginal_likelihood(th)
    t5 = time.perf_counter()
    print(f"N={n}: fit {(t3-t2)/30*1e3:.3f} ms, LML {(t5-t4)/20*1e3:.3f} ms  (alpha digest {float(gp.alpha.sum()).hex()}, lml {v.hex()})")
    gp.engine.close()
