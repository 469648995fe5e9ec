"""Constructor-with-search time (multi-start L-BFGS-B over the hyper-parameters, regression.py:585-605) for BASELINE
configs 1 and 4, lockstep (one batched gradient evaluation per round for all starts) against one start after another;
for the marginal likelihood and (round 4: gpmi_loo_grad_batch) the cross-validation objective (cross_val=True,
regression.py:159-164); and for a two-region ChangePoint model (round 4: gpmi_lml_grad_batch_mix).
usage: python tools/search_time.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor

out = []
ONLY = os.environ.get("SEARCH_ONLY", "")  # "cp": the ChangePoint cases only; "cp2048": the N = 2048 one only
for cfg, n, d in ((1, 512, 2), (5, 2048, 4), (4, 4096, 4)) if not ONLY else ():
    x, y, e = wl.synthetic_dataset(cfg, n, d)
    row = {"config": cfg, "N": n, "d": d}
    GpRegressor(x, y, y_err=e, hyperpars=wl.timing_theta(wl.SE, y, d)).marginal_likelihood_gradient_batch(
        np.array([wl.timing_theta(wl.SE, y, d)] * 5))  # warm-up: library, workspaces
    keep = GpRegressor._lockstep_search
    for mode in ("serial", "lockstep"):
        np.random.seed(3)
        t0 = time.perf_counter()
        GpRegressor._lockstep_search = (lambda self: False) if mode == "serial" else keep
        gp = GpRegressor(x, y, y_err=e)
        row[mode + "_seconds"] = time.perf_counter() - t0
        row[mode + "_lml"] = float(gp.marginal_likelihood(gp.hyperpars))
        row["starts"] = int(2 * np.sqrt(len(gp.hp_bounds))) + 1
    for mode in ("serial", "lockstep"):
        np.random.seed(3)
        GpRegressor._lockstep_search = (lambda self: False) if mode == "serial" else keep
        if mode == "serial":  # warm-up of the cross-validation path
            GpRegressor(x, y, y_err=e, hyperpars=wl.timing_theta(wl.SE, y, d), cross_val=True).loo_likelihood_gradient_batch(
                np.array([wl.timing_theta(wl.SE, y, d)] * 5))
        t0 = time.perf_counter()
        gp = GpRegressor(x, y, y_err=e, cross_val=True)
        row["cross_val_" + mode + "_seconds"] = time.perf_counter() - t0
        row["cross_val_" + mode + "_loo"] = float(gp.loo_likelihood(gp.hyperpars))
    GpRegressor._lockstep_search = keep
    out.append(row)
# a two-region ChangePoint model (covariance.py:371-606; round 4: gpmi_lml_grad_batch_mix), 1-D step in the length scale
from inference_amd.gp import ChangePoint, SquaredExponential

rng = np.random.default_rng(11)
for n in (512, 2048):
    if ONLY == "cp2048" and n != 2048:
        rng.uniform(0, 1, n), rng.normal(size=n)  # (the same data as in the full run)
        continue
    x = np.sort(rng.uniform(0, 1, n)).reshape(-1, 1)
    y = np.where(x[:, 0] < 0.5, np.sin(4 * x[:, 0]), np.sin(40 * x[:, 0])) + 0.05 * rng.normal(size=n)
    e = np.full(n, 0.05)
    row = {"config": "ChangePoint[SE, SE]", "N": n, "d": 1}
    keep = GpRegressor._lockstep_search
    for mode in ("warm", "serial", "lockstep"):
        np.random.seed(3)
        GpRegressor._lockstep_search = (lambda self: False) if mode == "serial" else keep
        t0 = time.perf_counter()
        gp = GpRegressor(x, y, y_err=e, kernel=ChangePoint(kernels=[SquaredExponential] * 2))
        if mode != "warm":
            row[mode + "_seconds"] = time.perf_counter() - t0
            row[mode + "_lml"] = float(gp.marginal_likelihood(gp.hyperpars))
            row["starts"] = int(2 * np.sqrt(len(gp.hp_bounds))) + 1
    GpRegressor._lockstep_search = keep
    out.append(row)
# differential evolution (regression.py:569-573): SciPy's default walk, one evaluation per call, against the opt-in batched
# form (round 6: diffev_batched=True, one lockstep call per generation), BASELINE config 1 and the N = 2048 shape
if not ONLY:
    for cfg, n, d in ((1, 512, 2), (5, 2048, 4)):
        x, y, e = wl.synthetic_dataset(cfg, n, d)
        row = {"config": cfg, "N": n, "d": d, "optimizer": "diffev"}
        for mode, kw in (("serial", {}), ("batched", {"diffev_batched": True})):
            np.random.seed(5)
            t0 = time.perf_counter()
            gp = GpRegressor(x, y, y_err=e, optimizer="diffev", **kw)
            row[f"diffev_{mode}_seconds"] = time.perf_counter() - t0
            row[f"diffev_{mode}_lml"] = float(gp.marginal_likelihood(gp.hyperpars))
        out.append(row)
print(json.dumps(out, indent=1))
# the accelerated path must never be the slower one (round 5's profile held a 2.55 s lockstep search beside a 0.94 s serial
# one for a whole round and nobody looked: profiles/r06_search_regression.txt)
slow = [(r["config"], r["N"], k) for r in out for k in ("", "cross_val_")
        if k + "lockstep_seconds" in r and r[k + "lockstep_seconds"] > r[k + "serial_seconds"]]
slow += [(r["config"], r["N"], "diffev") for r in out if "diffev_batched_seconds" in r
         and r["diffev_batched_seconds"] > r["diffev_serial_seconds"]]
if slow:
    print("search_time.py: the lockstep search is slower than the serial one for", slow, file=sys.stderr)
    sys.exit(1)
