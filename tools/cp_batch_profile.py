"""One lockstep gradient batch of the two-region ChangePoint model, repeated (for rocprofv3 --kernel-trace --stats and for
host-side timing): usage: python tools/cp_batch_profile.py [N] [B] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
from inference_amd.gp import GpRegressor, ChangePoint, SquaredExponential
n, B, reps = (int(a) for a in (sys.argv[1:] + ["2048", "6", "50"][len(sys.argv) - 1:]))
rng = np.random.default_rng(11)
x = np.sort(rng.uniform(0, 1, n)).reshape(-1, 1)
y = np.where(x[:, 0] < 0.5, np.sin(4 * x[:, 0]), np.sin(40 * x[:, 0])) + 0.05 * rng.normal(size=n)
e = np.full(n, 0.05)
th = np.array([0.1, -0.3, np.log(0.3), 0.2, np.log(0.04), 0.5, 0.05])
gp = GpRegressor(x, y, y_err=e, kernel=ChangePoint(kernels=[SquaredExponential] * 2), hyperpars=th)
gp.batch_independent_values(True)
X = th + 0.02 * rng.standard_normal((B, th.size))
X[:, -1] = np.abs(X[:, -1])
for _ in range(5):
    gp.marginal_likelihood_gradient_batch(X)
t0 = time.perf_counter()
for _ in range(reps):
    gp.marginal_likelihood_gradient_batch(X)
dt = (time.perf_counter() - t0) / reps
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(reps):
    gp.marginal_likelihood_gradient_batch(X)
pr.disable()
print(f"N={n} B={B}: {dt * 1e3:.3f} ms per batch call")
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
gp.engine.close()
