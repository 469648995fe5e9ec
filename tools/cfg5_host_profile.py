"""Host-side profile of BASELINE config 5 through the sharded driver as bench.py runs it (64 ladders x 8 temperatures,
10 steps, swap interval 10): where the Python time goes beside the device's.  usage: python tools/cfg5_host_profile.py [ladders] [steps]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd import sharding
from inference_amd.gp import GpRegressor
n_lad = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
x, y, e = wl.synthetic_dataset(5, 2048, 4)
gp = GpRegressor(x, y, y_err=e, hyperpars=wl.timing_theta(wl.SE, y, 4))
gp.batch_independent_values(True)
sharding.tempering_run(lambda k: wl.cfg5_ladder(gp, k), 1, 2, swap_interval=2)
t0 = time.perf_counter()
state, evals = sharding.tempering_run(lambda k: wl.cfg5_ladder(gp, k), n_lad, steps, swap_interval=10)
dt = time.perf_counter() - t0
print(f"{n_lad} ladders, {steps} steps: {evals} evaluations in {dt:.3f} s = {evals / dt:.0f} /s")
pr = cProfile.Profile()
pr.enable()
state, evals = sharding.tempering_run(lambda k: wl.cfg5_ladder(gp, k), n_lad, steps, swap_interval=10)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
