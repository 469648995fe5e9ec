"""Two consecutive fits + predicts at the headline size, results saved to an .npz
(tests/test_gpu_parity.py::test_schedule_variants_agree runs it in child processes under different scheduling
settings of the library and compares the results).
usage: python tools/fit_digest.py out.npz [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor

out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
d = 8
x, y, e = wl.synthetic_dataset(2, n, d)
theta = wl.timing_theta(wl.SE, y, d)
gp = GpRegressor(x, y, y_err=e, hyperpars=theta)
res = {}
for rep in range(2):  # twice: the second fit meets whatever the first one left on the streams
    gp.set_hyperparameters(theta)
    mu, sig = gp(wl.query_points(2, 256, d))
    res.update({f"alpha{rep}": gp.alpha.copy(), f"logdet{rep}": np.array([gp._logdet]), f"mu{rep}": mu, f"sig{rep}": sig})
np.savez(out, **res)
