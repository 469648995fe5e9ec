"""Is a lockstep value independent of the batch it is evaluated in?  For a two-region ChangePoint model (and a plain
SquaredExponential one) the pieces a gradient batch returns - LML, sub-kernel gradients, window row sums, alpha - for the
SAME hyper-parameter vector evaluated alone (batch of one), as row 0 of a batch of six and as row 5 of a batch of six, compared
bit for bit (the search iterates of the lockstep and the serial path fork in the 10th digit).
usage: python tools/batch_identity.py [N ...]"""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
from inference_amd.gp import GpRegressor, ChangePoint, SquaredExponential


def diff(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return {"max_abs": float(np.abs(a - b).max()), "scale": float(np.abs(b).max()), "equal": bool(np.array_equal(a, b))}


out = []
rng = np.random.default_rng(11)
for n in [int(a) for a in sys.argv[1:]] or [512, 2048]:
    x = np.sort(rng.uniform(0, 1, n)).reshape(-1, 1)
    y = np.where(x[:, 0] < 0.5, np.sin(4 * x[:, 0]), np.sin(40 * x[:, 0])) + 0.05 * rng.normal(size=n)
    e = np.full(n, 0.05)
    th = np.array([0.1, -0.3, np.log(0.3), 0.2, np.log(0.04), 0.5, 0.05])
    gp = GpRegressor(x, y, y_err=e, kernel=ChangePoint(kernels=[SquaredExponential] * 2), hyperpars=th)
    gp.batch_independent_values(True)  # a batch of one takes the lockstep kernels too (GPMI_OPT_LOCKSTEP_ALWAYS)
    others = th + 0.05 * rng.standard_normal((5, th.size))
    others[:, -1] = np.abs(others[:, -1])
    row = {"N": n, "model": "ChangePoint[SE, SE]"}

    def pieces(thetas):
        stat = [np.ascontiguousarray(t[gp.cov_slice][gp._stat_slice]) for t in thetas]
        args = [gp._mix_args(s_) for s_ in stat]
        win = [gp._mix_window_terms(s_) for s_ in stat]
        return gp.engine.lml_grad_batch_mix(args[0][0], [a[1] for a in args], np.array([a[2] for a in args]),
                                            np.zeros(len(thetas)), row_weights=np.array([w_[0] for w_ in win]),
                                            mu_const=np.array([t[0] for t in thetas]))

    one = pieces(np.array([th]))
    first = pieces(np.vstack([th[None, :], others]))
    last = pieces(np.vstack([others, th[None, :]]))
    for name, k in (("lml", 0), ("sub_kernel_gradients", 1), ("window_row_sums", 2), ("alpha", 3)):
        row[name] = {"alone_vs_row0_of_6": diff(one[k][0], first[k][0]), "alone_vs_row5_of_6": diff(one[k][0], last[k][-1])}
    f1, g1 = gp.marginal_likelihood_gradient(th)  # (the single-evaluation entry point: stream / flag-ordered factorisation)
    fb, gb = gp.marginal_likelihood_gradient_batch(np.vstack([th[None, :], others]))
    row["single_evaluation_path_vs_row0_of_6"] = {"lml": diff(f1, fb[0]), "gradient": diff(g1, gb[0])}
    out.append(row)
    gp.engine.close()
    # the plain kernel's batch for comparison
    gp = GpRegressor(x, y, y_err=e, kernel=SquaredExponential, hyperpars=th[:3])
    gp.batch_independent_values(True)
    o3 = th[:3] + 0.05 * rng.standard_normal((5, 3))
    f1, g1 = gp.marginal_likelihood_gradient_batch(th[None, :3])
    fb, gb = gp.marginal_likelihood_gradient_batch(np.vstack([th[None, :3], o3]))
    fl, gl = gp.marginal_likelihood_gradient_batch(np.vstack([o3, th[None, :3]]))
    out.append({"N": n, "model": "SquaredExponential", "lml": {"alone_vs_row0_of_6": diff(f1[0], fb[0]), "alone_vs_row5_of_6": diff(f1[0], fl[-1])},
                "gradient": {"alone_vs_row0_of_6": diff(g1[0], gb[0]), "alone_vs_row5_of_6": diff(g1[0], gl[-1])}})
    gp.engine.close()
print(json.dumps(out, indent=1))
