"""Where one GpOptimiser.propose_evaluation (config 4: N = 4096, d = 4, 4096 L-BFGS-B starts in lockstep) spends its wall
time: inside the batched acquisition calls (device + wrappers) vs the host-side L-BFGS-B driver around them."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import ExpectedImprovement, GpOptimiser
n, d = 4096, 4
x, y, e = wl.synthetic_dataset(4, n, d)
opt = GpOptimiser(x, y, bounds=[(0.0, 1.0)] * d, y_err=e, hyperpars=wl.timing_theta(wl.SE, y, d), acquisition=ExpectedImprovement)
acc = {"g": 0.0, "gn": 0, "gp": 0, "f": 0.0, "fn": 0, "fp": 0}
og, of = opt.acquisition.opt_func_gradient_batch, getattr(opt.acquisition, "opt_func_batch", None)
def cg(p):
    t0 = time.perf_counter(); r = og(p); acc["g"] += time.perf_counter() - t0; acc["gn"] += 1; acc["gp"] += len(p); return r
opt.acquisition.opt_func_gradient_batch = cg
if of is not None:
    def cf(p):
        t0 = time.perf_counter(); r = of(p); acc["f"] += time.perf_counter() - t0; acc["fn"] += 1; acc["fp"] += len(p); return r
    opt.acquisition.opt_func_batch = cf
np.random.seed(1); opt.propose_evaluation()
for k in acc: acc[k] = 0
np.random.seed(1)
t0 = time.perf_counter(); opt.propose_evaluation(); dt = time.perf_counter() - t0
print(f"propose_evaluation {dt*1e3:.0f} ms: gradient batches {acc['g']*1e3:.0f} ms in {acc['gn']} calls ({acc['gp']} points), value batches {acc['f']*1e3:.0f} ms in {acc['fn']} calls ({acc['fp']} points), everything else {1e3*(dt-acc['g']-acc['f']):.0f} ms")
pr = cProfile.Profile(); np.random.seed(1); pr.enable(); opt.propose_evaluation(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
