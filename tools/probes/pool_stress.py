"""(Record of the experiment in profiles/HISTORY.md R6.18: the GPMI_PAIR_MIN_TILES / GPMI_PAIR_POOL_SMALL switches and the pool it
exercised were a temporary patch of api.hip and are not in the library.)
Stress of the process-wide pool of CU-masked stream pairs: many handles opened and closed one after another (the pool's one
pair is reused), several alive at once (the pool grows), values checked against the first; then a clean exit (gpmi_shutdown
from the binding's atexit hook).  usage: python tools/probes/pool_stress.py [cycles]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n, d = 2048, 4
x, y, e = wl.synthetic_dataset(4, n, d)
th = wl.timing_theta(wl.SE, y, d)
ref = None
t0 = time.perf_counter()
for k in range(cycles):
    gp = GpRegressor(x, y, y_err=e, hyperpars=th)
    v = gp.marginal_likelihood(th)
    a = gp.alpha.copy()
    if ref is None: ref = (v, a)
    assert v == ref[0] and np.array_equal(a, ref[1]), k
    gp.engine.close()
dt = (time.perf_counter() - t0) / cycles
print(f"{cycles} handles one after another at N={n}: {dt*1e3:.1f} ms per construct + LML + close, values identical")
alive = [GpRegressor(x, y, y_err=e, hyperpars=th) for _ in range(4)]
for r in range(3):
    for gp in alive:
        gp.set_hyperparameters(th)
        assert np.array_equal(gp.alpha, ref[1])
        assert gp.marginal_likelihood(th) == ref[0]
print("4 handles alive at once, 3 rounds of fit + LML each: values identical")
for gp in alive[:2]: gp.engine.close()
gp = GpRegressor(x, y, y_err=e, hyperpars=th); assert np.array_equal(gp.alpha, ref[1])
big = GpRegressor(*wl.synthetic_dataset(2, 8192, 8)[:2], y_err=wl.synthetic_dataset(2, 8192, 8)[2], hyperpars=wl.timing_theta(wl.SE, wl.synthetic_dataset(2, 8192, 8)[1], 8))
t0 = time.perf_counter()
for _ in range(20): big.set_hyperparameters(big.hyperpars)
print(f"N=8192 fit beside them: {(time.perf_counter()-t0)/20*1e3:.3f} ms")
print("leaving some handles open: the atexit hook closes them and shuts the pool down")
