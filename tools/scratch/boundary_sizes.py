"""Fits at sizes around the look-ahead / flag-ordered-tail switch (52 trailing tile rows after the first outer panel: 56 tile
rows = N 7168) checked through K alpha = y - mu and log det against the host (oracle kernel, LAPACK)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor, SquaredExponential
from oracle import gp_oracle as orc

worst = 0.0
for n in (6528, 6900, 7040, 7168, 7169, 7296, 7700, 8320, 9100):
    d = 4
    x, y, e = wl.synthetic_dataset(n, n, d)
    th = wl.timing_theta(wl.SE, y, d)
    gp = GpRegressor(x, y, y_err=e, hyperpars=th, kernel=SquaredExponential())
    K = orc.se_build(x, th[1:])
    K[np.diag_indices(n)] += e**2
    r = K @ gp.alpha - (y - th[0])
    err = np.abs(r).max() / np.abs(y - th[0]).max()
    L = np.linalg.cholesky(K)
    ld = np.log(np.diag(L)).sum()
    e2 = abs(gp._logdet - ld) / abs(ld)
    worst = max(worst, err, e2)
    print(f"N={n}: |K alpha - (y - mu)| / |y - mu| = {err:.2e}, log det {e2:.2e}")
print("worst", worst)
assert worst < 1e-9
