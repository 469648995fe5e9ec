"""One rank of the two-processes-on-one-device test (tests/test_gpu_parity.py::test_two_ranks_on_one_device): runs the
three sharded drivers on DEVICE engines through a FileRendezvous (RCCL refuses two ranks on one device, so the gather
takes the rendezvous files - the path the first multi-GPU run falls back to) and saves what every rank must agree on.
usage: python tests/sharded_rank.py out.npz   (RANK / WORLD_SIZE / GPMI_RDV_KEY in the environment)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np  # noqa: E402
import workloads as wl  # noqa: E402
from inference_amd import sharding  # noqa: E402
from inference_amd.gp import GpRegressor, RationalQuadratic  # noqa: E402

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
rdv = sharding.FileRendezvous(rank, world) if world > 1 else None
sharding.use_rendezvous(rdv)
out = {}
try:
    # config 3 shape: RQ, d = 16, a 13-point grid (uneven blocks)
    x, y, e = wl.synthetic_dataset(3, 1200, 16)
    grid = wl.theta_grid_cfg3(y, 16)[:13]
    gp = GpRegressor(x, y, y_err=e, hyperpars=grid[0], kernel=RationalQuadratic)
    gp.engine.set_streams(2)
    gp.batch_independent_values(True)  # a value must not depend on the size of the batch it is evaluated in
    out["sweep"] = sharding.marginal_likelihood_sweep(gp, grid)
    # multi-start search: 5 starts
    x1, y1, e1 = wl.synthetic_dataset(1, 300, 2)
    gp1 = GpRegressor(x1, y1, y_err=e1, hyperpars=wl.timing_theta(wl.SE, y1, 2))
    starts = wl.theta_set(wl.SE, y1, 2, 5, seed=3)
    th, f = sharding.multistart_sweep(gp1, starts)
    out["ms_theta"], out["ms_f"] = th, f
    # config 5 shape: 3 ladders x 8 temperatures, N = 512
    x5, y5, e5 = wl.synthetic_dataset(5, 512, 4)
    gp5 = GpRegressor(x5, y5, y_err=e5, hyperpars=wl.timing_theta(wl.SE, y5, 4))
    gp5.batch_independent_values(True)
    state, evals = sharding.tempering_run(lambda k: wl.cfg5_ladder(gp5, k), 3, 6, swap_interval=3)
    out["pt_state"], out["pt_evals"] = state, np.array([evals])
    np.savez(sys.argv[1], **out)
finally:
    if rdv is not None:
        rdv.close()
