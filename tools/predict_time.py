"""Prediction time (GpRegressor.__call__, regression.py:168-216) at the headline's and config 2's sizes.
usage: python tools/predict_time.py [N ...]   (M = 256, 1024, 2048 query points each)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor

for n in [int(a) for a in sys.argv[1:]] or [16384, 8192]:
    x, y, e = wl.synthetic_dataset(1, n, 8)
    th = wl.timing_theta(wl.SE, y, 8)
    gp = GpRegressor(x, y, y_err=e, hyperpars=th)
    out = []
    for m in (256, 1024, 2048):
        pts = wl.query_points(1, m, 8)
        gp(pts)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            mu, sig = gp(pts)
            ts.append(time.perf_counter() - t0)
        dt = min(ts)
        out.append(f"M={m} {dt*1e3:.2f} ms ({m * n * n / dt / 1e12:.1f} TFLOP/s, digest {float(np.sum(mu) + np.sum(sig)):.17g})")
    print(f"N={n}: " + " | ".join(out))
