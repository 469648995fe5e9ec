"""How the time of a public call depends on how long the device has been kept busy (the shader clock ramps up over tens of
milliseconds of sustained load: tools/probes/mfma_sustained.hip, tools/bench_gemm.py with 5 against 40 repetitions).
usage: python tools/clock_ramp.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor

x, y, e = wl.synthetic_dataset(2, 8192, 8)
th = wl.timing_theta(wl.SE, y, 8)
pts = wl.query_points(2, 1024, 8)
gp = GpRegressor(x, y, y_err=e, hyperpars=th)
for name, fn in (("fit", lambda: gp.set_hyperparameters(th)), ("predict", lambda: gp(pts)), ("lml", lambda: gp.marginal_likelihood(th))):
    row = []
    for reps in (1, 3, 10, 30, 100):
        time.sleep(0.5)  # let the clock fall back
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        row.append(f"{reps} calls: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms")
    # per-call times inside one long run
    time.sleep(0.5)
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"N=8192 {name}: " + " | ".join(row) + f" | call by call after 0.5 s idle: first {ts[0]:.2f}, 2nd {ts[1]:.2f}, 5th {ts[4]:.2f}, 10th {ts[9]:.2f}, 20th {ts[19]:.2f}, 40th {ts[39]:.2f} ms")
