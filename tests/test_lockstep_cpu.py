"""The lockstep L-BFGS-B driver behind GpOptimiser.multistart_bfgs reproduces scipy.optimize.fmin_l_bfgs_b run by run
(iterates, objective values, call and iteration counts) on an analytic bounded problem - CPU only."""
import numpy as np
from scipy.optimize import fmin_l_bfgs_b

from inference_amd.gp._lockstep import lockstep_lbfgsb


def _problem():
    rng = np.random.default_rng(0)
    A = rng.standard_normal((3, 3))
    A = A @ A.T + np.eye(3)

    def batch(X):
        return (np.array([0.5 * x @ A @ x + np.sin(x).sum() for x in X]), np.array([A @ x + np.cos(x) for x in X]))

    return batch, [(-1, 2), (-0.5, None), (None, 0.3)], rng.uniform(-1, 1, (9, 3))


def test_lockstep_runs_equal_serial_fmin_l_bfgs_b():
    batch, bounds, starts = _problem()
    calls = []

    def counted(X):
        calls.append(len(X))
        return batch(X)

    res = lockstep_lbfgsb(counted, starts, bounds, pgtol=1e-10)
    for x0, (x, f, d) in zip(starts, res):
        xr, fr, dr = fmin_l_bfgs_b(lambda v: tuple(a[0] for a in batch(v[None, :])), x0, approx_grad=False,
                                   bounds=bounds, pgtol=1e-10)
        assert np.array_equal(x, xr) and f == fr
        assert (d["funcalls"], d["nit"], d["warnflag"]) == (dr["funcalls"], dr["nit"], dr["warnflag"])
    # one batched call per round: as many calls as the longest run needs, not the sum over the runs
    assert len(calls) == max(r[2]["funcalls"] for r in res) and calls[0] == len(starts)
    assert sum(calls) == sum(r[2]["funcalls"] for r in res)


def test_lockstep_falls_back_when_the_private_entry_point_differs(monkeypatch):
    """Another SciPy whose `setulb` has a different signature (TypeError): same results from the serial fall-back."""
    from inference_amd.gp import _lockstep

    batch, bounds, starts = _problem()
    want = lockstep_lbfgsb(batch, starts[:3], bounds, pgtol=1e-10)

    def other_signature(*a, **k):
        raise TypeError("setulb() takes 12 positional arguments")

    monkeypatch.setattr(_lockstep, "_drive", other_signature)
    got = lockstep_lbfgsb(batch, starts[:3], bounds, pgtol=1e-10)
    for a, b in zip(want, got):
        assert np.array_equal(a[0], b[0]) and a[1] == b[1]
