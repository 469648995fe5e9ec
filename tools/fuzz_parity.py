"""Randomised parity sweep against the CPU oracle (debug aid, not part of the suites): random sizes, dimensions,
kernels, noise levels; fit, LML, LML gradient, predict."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor, SquaredExponential, RationalQuadratic, WhiteNoise
from oracle import gp_oracle as orc

def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 16):
    n = int(rng.integers(2, 2600)); d = int(rng.integers(1, 12)); kid = int(rng.integers(0, 2)); wn = bool(rng.integers(0, 2))
    x, y, e = wl.synthetic_dataset(1000 + case, n, d)
    e = e * float(rng.uniform(0.5, 3.0))
    th = wl.timing_theta(kid, y, d) + 0.2 * rng.standard_normal(wl.timing_theta(kid, y, d).size)
    cov = (SquaredExponential if kid == wl.SE else RationalQuadratic)()
    if wn:
        cov = cov + WhiteNoise(); th = np.append(th, np.log(0.05))
    gp = GpRegressor(x, y, y_err=e, hyperpars=th, kernel=cov)
    ref = orc.OracleGp(x, y, e, kernel=kid, hyperpars=th, white_noise=wn)
    pts = wl.query_points(case, 33, d)
    mu, sig = gp(pts); rmu, rsig = ref(pts)
    l, g = gp.marginal_likelihood_gradient(th); rl, rg = ref.marginal_likelihood_gradient(th)
    errs = dict(alpha=rel(gp.alpha, ref.alpha), mu=rel(mu, rmu), sig=rel(sig, rsig), lml=rel(l, rl), grad=rel(g, rg))
    w = max(errs.values()); worst = max(worst, w)
    print(f"n={n:5d} d={d:2d} kernel={'SE' if kid == 0 else 'RQ'}{'+WN' if wn else '   '}  worst {w:.2e}  " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()))
print("overall worst relative error", worst)
