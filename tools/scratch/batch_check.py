"""Lockstep batches of several sizes (a multiple of 8 problems per half-batch runs one problem per XCD) against one-at-a-time
evaluations: LML values and gradients must agree to rounding, whatever the batch size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor, SquaredExponential

rng = np.random.default_rng(3)
for n, d in ((700, 3), (2048, 4), (3000, 5)):
    x, y, e = wl.synthetic_dataset(7, n, d)
    th0 = wl.timing_theta(wl.SE, y, d)
    gp = GpRegressor(x, y, y_err=e, hyperpars=th0, kernel=SquaredExponential())
    for T in (3, 8, 16, 17, 32, 64):
        ths = th0 + 0.1 * rng.standard_normal((T, th0.size))
        lb = gp.marginal_likelihood_batch(ths)
        ls = np.array([gp.marginal_likelihood(t) for t in ths[: min(T, 6)]])
        vb, gb = gp.marginal_likelihood_gradient_batch(ths[: min(T, 16)])
        vs = [gp.marginal_likelihood_gradient(t) for t in ths[:3]]
        e1 = np.abs(lb[: len(ls)] - ls).max() / np.abs(ls).max()
        e2 = max(np.abs(gb[i] - vs[i][1]).max() / np.abs(vs[i][1]).max() for i in range(3))
        print(f"n={n} T={T}: LML batch vs single {e1:.1e}, gradient batch vs single {e2:.1e}")
        assert e1 < 1e-11 and e2 < 1e-9
print("ok")
