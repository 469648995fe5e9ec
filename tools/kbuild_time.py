"""Covariance-build time in isolation (HIP events around the build class): SE d = 8 and RQ d = 16 at N = 16384, and the
batched build of config 5.  usage: python tools/kbuild_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd import _lib
from inference_amd.gp import GpRegressor, RationalQuadratic

for kid, d, cfg, kern in ((wl.SE, 8, 2, None), (wl.RQ, 16, 3, RationalQuadratic)):
    x, y, e = wl.synthetic_dataset(cfg, 16384, d)
    th = wl.timing_theta(kid, y, d)
    gp = GpRegressor(x, y, y_err=e, hyperpars=th, **({"kernel": kern} if kern else {}))
    eng = gp.engine
    eng.profile_enable(2 << _lib.PROF_KBUILD)
    eng.profile_reset()
    for _ in range(5):
        gp.set_hyperparameters(th)
    eng.sync()
    p = eng.profile_read(_lib.PROF_KBUILD)
    eng.profile_enable(0)
    ms = p["ms"] / max(p["launches"], 1)
    print(f"{'SE' if kid == wl.SE else 'RQ'} d={d} N=16384 lower-tile build: {ms:.3f} ms per build = {p['bytes'] / p['launches'] / ms / 1e9:.2f} TB/s "
          f"({p['bytes'] / p['launches'] / ms / 1e9 / 8.0 * 100:.0f} % of 8 TB/s), {p['launches']} builds")
    t0 = time.perf_counter(); gp.marginal_likelihood(th); gp.marginal_likelihood(th); t1 = time.perf_counter()
    print(f"   LML {(t1 - t0) / 2 * 1e3:.1f} ms")
