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
    """Another SciPy: (a) a `setulb` with a different signature (raises), (b) one that runs but behaves differently (same
    arity, other numbers) - the self-check rejects both and the serial fall-back gives the same results."""
    from inference_amd.gp import _lockstep

    batch, bounds, starts = _problem()
    want = lockstep_lbfgsb(batch, starts[:3], bounds, pgtol=1e-10)
    assert _lockstep.DRIVER_STATE["ok"], _lockstep.DRIVER_STATE["why"]
    real_drive = _lockstep._drive

    def other_signature(*a, **k):
        raise TypeError("setulb() takes 12 positional arguments")

    def other_numbers(setulb, fun_batch, starts_, *a, **k):
        out = real_drive(setulb, fun_batch, starts_, *a, **k)
        return [(x * (1 + 1e-9), f, d) for x, f, d in out]

    for fake in (other_signature, other_numbers):
        monkeypatch.setattr(_lockstep, "_drive", fake)
        monkeypatch.setattr(_lockstep, "DRIVER_STATE", {"checked": False, "ok": False, "why": ""})
        got = lockstep_lbfgsb(batch, starts[:3], bounds, pgtol=1e-10)
        assert not _lockstep.DRIVER_STATE["ok"] and _lockstep.DRIVER_STATE["why"]
        for a_, b_ in zip(want, got):
            assert np.array_equal(a_[0], b_[0]) and a_[1] == b_[1]
