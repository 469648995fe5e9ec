"""BASELINE config 5 on one GPU: 8 ParallelTempering ladders x 8 temperatures over the GP log-marginal likelihood
(N = 2048, d = 4), advanced in lockstep.  Prints one JSON line: LML evaluations/s, chain steps/s.
usage: python tools/config5_bench.py [steps] [ladders]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np  # noqa: E402
import workloads as wl  # noqa: E402
from inference_amd.gp import GpRegressor  # noqa: E402
from inference_amd.mcmc import advance_ladders  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_lad = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n, d = 2048, 4
x, y, e = wl.synthetic_dataset(5, n, d)
gp = GpRegressor(x, y, y_err=e, hyperpars=wl.timing_theta(wl.SE, y, d))
gp.batch_independent_values(True)
ladders = [wl.cfg5_ladder(gp, k) for k in range(n_lad)]
advance_ladders(ladders, 2, swap_interval=2)  # warm-up
t0 = time.perf_counter()
evals = advance_ladders(ladders, steps, swap_interval=10)
dt = time.perf_counter() - t0
chains = sum(len(l.chains) for l in ladders)
print(json.dumps({"config": f"{n_lad} ladders x 8 temperatures, GibbsChain over P={gp.n_hyperpars} hyper-parameters, SE N={n} d={d}",
                  "steps": steps, "lml_evaluations": evals, "seconds": dt, "lml_evals_per_s": evals / dt,
                  "chain_steps_per_s": chains * steps / dt, "tflops": evals / dt * (n**3 / 3.0) / 1e12,
                  "swaps_accepted": int(sum(l.successful_swaps.sum() for l in ladders))}))
