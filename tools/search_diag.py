"""What a constructor-with-search of a two-region ChangePoint model does, call by call:
rounds, evaluations per round, seconds per device call, for the lockstep and the serial multi-start search
(regression.py:585-605, covariance.py:546-594).  Works against any tree that has inference_amd.gp (pass its root):
usage: python tools/search_diag.py [N ...] [--root TREE]"""
import json, os, sys, time

args = sys.argv[1:]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--root" in args:
    i = args.index("--root")
    ROOT = os.path.abspath(args[i + 1])
    del args[i:i + 2]
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
from inference_amd.gp import GpRegressor, ChangePoint, SquaredExponential

sizes = [int(a) for a in args] or [512, 2048]
rng = np.random.default_rng(11)
out = []
for n in sizes:
    x = np.sort(rng.uniform(0, 1, n)).reshape(-1, 1)
    y = np.where(x[:, 0] < 0.5, np.sin(4 * x[:, 0]), np.sin(40 * x[:, 0])) + 0.05 * rng.normal(size=n)
    e = np.full(n, 0.05)
    keep = GpRegressor._lockstep_search
    keep_b = GpRegressor.marginal_likelihood_gradient_batch
    keep_s = GpRegressor.marginal_likelihood_gradient
    row = {"N": n}
    for mode in ("warm", "serial", "lockstep"):
        calls = []

        def batch(self, thetas, _calls=calls):
            t0 = time.perf_counter()
            r = keep_b(self, thetas)
            _calls.append((len(np.atleast_2d(thetas)), time.perf_counter() - t0))
            if os.environ.get("DIAG_SLOW_MS") and _calls[-1][1] * 1e3 > float(os.environ["DIAG_SLOW_MS"]):
                print("slow call", len(_calls), round(_calls[-1][1] * 1e3, 2), "ms  thetas:", np.atleast_2d(thetas).tolist(),
                      "lml:", np.asarray(r[0]).tolist(), file=sys.stderr)
            return r

        def single(self, theta, _calls=calls):
            t0 = time.perf_counter()
            r = keep_s(self, theta)
            _calls.append((1, time.perf_counter() - t0))
            return r

        GpRegressor.marginal_likelihood_gradient_batch = batch
        GpRegressor.marginal_likelihood_gradient = single
        np.random.seed(3)
        GpRegressor._lockstep_search = (lambda self: False) if mode == "serial" else keep
        t0 = time.perf_counter()
        gp = GpRegressor(x, y, y_err=e, kernel=ChangePoint(kernels=[SquaredExponential] * 2))
        dt = time.perf_counter() - t0
        GpRegressor.marginal_likelihood_gradient_batch = keep_b
        GpRegressor.marginal_likelihood_gradient = keep_s
        if mode == "warm":
            continue
        secs = np.array([c[1] for c in calls])
        sizes_ = np.array([c[0] for c in calls])
        row[mode] = {
            "seconds": dt, "device_calls": len(calls), "evaluations": int(sizes_.sum()),
            "in_calls_seconds": float(secs.sum()), "median_call_ms": float(np.median(secs) * 1e3),
            "max_call_ms": float(secs.max() * 1e3), "slowest_calls_ms": [round(float(s) * 1e3, 2) for s in np.sort(secs)[-5:]],
            "batch_sizes_first_rounds": [int(s) for s in sizes_[:12]],
            "rounds_by_batch_size": {str(int(b)): int((sizes_ == b).sum()) for b in np.unique(sizes_)},
            "lml": float(gp.marginal_likelihood(gp.hyperpars)),
            "call_ms_series": [round(float(s) * 1e3, 1) for s in secs] if os.environ.get("DIAG_SERIES") else None,
            "search_log_f": [float(s[2]) for s in getattr(gp, "search_log", [])],
        }
    GpRegressor._lockstep_search = keep
    # one theta: single evaluation against a batch of one and against its value inside a batch of six
    th = np.array(gp.hyperpars)
    l1, g1 = gp.marginal_likelihood_gradient(th)
    lb, gb = gp.marginal_likelihood_gradient_batch(th[None, :])
    l6, g6 = gp.marginal_likelihood_gradient_batch(np.tile(th, (6, 1)) + np.arange(6)[:, None] * 1e-3)
    row["single_vs_batch1"] = {"lml_diff": float(abs(l1 - lb[0])), "grad_maxdiff": float(np.abs(g1 - gb[0]).max()),
                               "grad_scale": float(np.abs(g1).max()),
                               "batch1_vs_in_batch6_lml_diff": float(abs(lb[0] - l6[0])),
                               "batch1_vs_in_batch6_grad_maxdiff": float(np.abs(gb[0] - g6[0]).max())}
    out.append(row)
print(json.dumps(out, indent=1))
