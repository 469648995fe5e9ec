"""BASELINE config 4 as the optimiser sees it: one GpOptimiser.propose_evaluation() on N = 4096 training points in d = 4
(fixed hyper-parameters): 4096 x 20 probe evaluations for the starting positions + 4096 L-BFGS-B runs in lockstep.
Prints one JSON line.  usage: python tools/propose_bench.py [n] [serial_subset]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np  # noqa: E402
import workloads as wl  # noqa: E402
from inference_amd.gp import ExpectedImprovement, GpOptimiser  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
subset = int(sys.argv[2]) if len(sys.argv) > 2 else 48
d = 4
x, y, e = wl.synthetic_dataset(4, n, d)
theta = wl.timing_theta(wl.SE, y, d)
bounds = [(0.0, 1.0)] * d
opt = GpOptimiser(x, y, bounds=bounds, y_err=e, hyperpars=theta, acquisition=ExpectedImprovement)
calls = {"n": 0, "pts": 0}
orig = opt.acquisition.opt_func_gradient_batch


def counted(p):
    calls["n"] += 1
    calls["pts"] += len(p)
    return orig(p)


opt.acquisition.opt_func_gradient_batch = counted
np.random.seed(1)
opt.propose_evaluation()  # warm-up (workspaces)
calls.update(n=0, pts=0)
np.random.seed(1)
t0 = time.perf_counter()
prop = opt.propose_evaluation()
dt = time.perf_counter() - t0
val = float(opt.acquisition.opt_func(prop))
# the reference's serial structure on a subset of the same starts: one fmin_l_bfgs_b per start, M = 1 device calls
np.random.seed(1)
starts = opt.acquisition.starting_positions(bounds)[:subset]
t0 = time.perf_counter()
serial = [opt.launch_bfgs(s) for s in starts]
dt_serial = time.perf_counter() - t0
from inference_amd.gp._lockstep import lockstep_lbfgsb  # noqa: E402

lock = lockstep_lbfgsb(orig, np.array(starts), bounds, pgtol=1e-10)
agree = max(abs(float(a[1]) - float(b[1])) / max(abs(float(a[1])), 1.0) for a, b in zip(serial, lock))
print(json.dumps({"config": f"GpOptimiser.propose_evaluation, EI, SE N={n} d={d}, {n} L-BFGS-B starts (pgtol 1e-10)",
                  "seconds": dt, "batched_gradient_calls": calls["n"], "points_evaluated": calls["pts"],
                  "proposal": [float(v) for v in np.ravel(prop)], "minus_ln_EI": val,
                  "serial_seconds_per_start": dt_serial / subset,
                  "serial_estimate_all_starts_s": dt_serial / subset * n,
                  "lockstep_vs_serial_objective_agreement": agree}))
