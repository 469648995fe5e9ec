"""
Many L-BFGS-B minimisations advanced in lockstep, one batched objective evaluation per round.

`GpOptimiser.multistart_bfgs` (reference: inference/gp/optimisation.py:202-223) starts one
`scipy.optimize.fmin_l_bfgs_b` run per training point; each of its iterations evaluates the acquisition
function and its gradient at ONE point, i.e. two M = 1 device calls (predict + spatial derivatives) that
cost what M = 1000 costs.  Here every run is the same SciPy L-BFGS-B (the Fortran-derived `setulb` routine
driven through its reverse-communication interface, exactly the loop of scipy.optimize._lbfgsb_py), but all
runs that are waiting for an objective value are served by one batched call `fun_batch(X) -> (f, G)`.
The iterates of a run are those of `fmin_l_bfgs_b` up to the last bits in which a batched and a single
device evaluation of the same point may differ.

The driver uses SciPy's private `_lbfgsb.setulb` (written against SciPy 1.15's `_minimize_lbfgsb`).  A SciPy whose
`setulb` differs can differ silently - same arity, other work-array sizes or task codes - so the driver is trusted
only after a self-check in this process: before its first use it minimises a small bounded test function through
the driver and through the public `fmin_l_bfgs_b` and requires identical iterates (x, f, function calls, iterations
bit for bit).  If the entry point is missing, raises, or fails that check, the runs fall back to `fmin_l_bfgs_b`
one after another on the same batched objective (same results, no lockstep) and `DRIVER_STATE` says why.
"""
import numpy as np
from scipy.optimize import fmin_l_bfgs_b


class _Run:
    """Reverse-communication state of one L-BFGS-B run (scipy/optimize/_lbfgsb_py.py, _minimize_lbfgsb)."""

    def __init__(self, x0, clip_low, clip_upp, m):
        n = x0.size
        self.x = np.array(np.clip(x0, clip_low, clip_upp), dtype=np.float64)
        self.f = np.array(0.0, dtype=np.float64)
        self.g = np.zeros(n, dtype=np.float64)
        self.wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
        self.iwa = np.zeros(3 * n, dtype=np.int32)
        self.task = np.zeros(2, dtype=np.int32)
        self.ln_task = np.zeros(2, dtype=np.int32)
        self.lsave = np.zeros(4, dtype=np.int32)
        self.isave = np.zeros(44, dtype=np.int32)
        self.dsave = np.zeros(29, dtype=np.float64)
        self.nfev = 0
        self.nit = 0
        self.done = False


def low_or(inf, bound, nbd, one_sided):
    """bound where the variable has that side bounded (nbd 2 = both, 1 = lower only, 3 = upper only), else +-inf."""
    return np.where((nbd == 2) | (nbd == one_sided), bound, inf)


def _encode_bounds(bounds, n):
    nbd = np.zeros(n, np.int32)
    low, upp = np.zeros(n), np.zeros(n)
    for i, (lo, hi) in enumerate(bounds):
        has_lo = lo is not None and np.isfinite(lo)
        has_hi = hi is not None and np.isfinite(hi)
        if has_lo:
            low[i] = lo
        if has_hi:
            upp[i] = hi
        nbd[i] = {(False, False): 0, (True, False): 1, (True, True): 2, (False, True): 3}[(has_lo, has_hi)]
    return low, upp, nbd


DRIVER_STATE = {"checked": False, "ok": False, "why": "not checked yet"}


def _self_check(setulb):
    """The reverse-communication driver against the public entry point on a bounded 4-D test problem (a tilted
    Rosenbrock chain with two active bounds): identical x, f, funcalls and nit, or the driver is not used."""

    def fg(x):
        f = np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1.0 - x[:-1]) ** 2) + 0.1 * x[-1]
        g = np.zeros_like(x)
        g[:-1] = -400.0 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2.0 * (1.0 - x[:-1])
        g[1:] += 200.0 * (x[1:] - x[:-1] ** 2)
        g[-1] += 0.1
        return float(f), g

    def batch(X):
        r = [fg(x) for x in X]
        return np.array([a for a, _ in r]), np.array([b for _, b in r])

    bounds = [(-1.5, 0.8), (None, 2.0), (-0.5, None), (None, None)]
    starts = np.array([[-1.2, 1.0, 0.3, -0.7], [0.5, 0.5, 0.5, 0.5], [0.8, -1.0, 2.0, 1.5]])
    kw = dict(pgtol=1e-9, factr=1e7, m=10, maxfun=15000, maxiter=15000, maxls=20)
    try:
        got = _drive(setulb, batch, starts, bounds, 4, **kw)
    except Exception as err:  # signature / dtype / shape mismatch of a different SciPy
        return False, f"driver raised {type(err).__name__}: {err}"
    for (x, f, d), x0 in zip(got, starts):
        xr, fr, dr = fmin_l_bfgs_b(fg, x0, approx_grad=False, bounds=bounds, **kw)
        if not (np.array_equal(x, xr) and f == float(fr) and d["funcalls"] == dr["funcalls"] and d["nit"] == dr["nit"]
                and d["warnflag"] == dr["warnflag"]):
            return False, "driver and fmin_l_bfgs_b disagree on the self-check problem"
    return True, "setulb driver verified against fmin_l_bfgs_b"


def _driver():
    """SciPy's setulb if this process has verified the driver against the public entry point, else None."""
    if not DRIVER_STATE["checked"]:
        DRIVER_STATE["checked"] = True
        try:
            from scipy.optimize import _lbfgsb

            setulb = _lbfgsb.setulb
        except (ImportError, AttributeError):
            DRIVER_STATE.update(ok=False, why="scipy.optimize._lbfgsb.setulb not found")
            return None
        ok, why = _self_check(setulb)
        DRIVER_STATE.update(ok=ok, why=why, setulb=setulb if ok else None)
    return DRIVER_STATE.get("setulb") if DRIVER_STATE["ok"] else None


def lockstep_lbfgsb(fun_batch, starts, bounds, pgtol=1e-5, factr=1e7, m=10, maxfun=15000, maxiter=15000, maxls=20):
    """Minimise from every row of `starts`; `fun_batch(X (B, n)) -> (f (B,), G (B, n))`.
    Returns a list of `(x, f, info)` in the order of `starts`, `info` with the keys of fmin_l_bfgs_b's dict
    ('warnflag', 'funcalls', 'nit', 'grad')."""
    starts = np.atleast_2d(np.asarray(starts, dtype=float))
    n = starts.shape[1]
    setulb = _driver()
    if setulb is not None:
        return _drive(setulb, fun_batch, starts, bounds, n, pgtol, factr, m, maxfun, maxiter, maxls)

    def single(x):
        f, g = fun_batch(np.asarray(x, dtype=float)[None, :])
        return float(f[0]), np.asarray(g[0], dtype=float)

    out = []
    for x0 in starts:
        x, f, d = fmin_l_bfgs_b(single, x0, approx_grad=False, bounds=bounds, pgtol=pgtol, factr=factr, m=m,
                                maxfun=maxfun, maxiter=maxiter, maxls=maxls)
        out.append((x, float(f), d))
    return out


def _drive(setulb, fun_batch, starts, bounds, n, pgtol, factr, m, maxfun, maxiter, maxls):
    low, upp, nbd = _encode_bounds(bounds, n)
    clip_low, clip_upp = low_or(-np.inf, low, nbd, 1), low_or(np.inf, upp, nbd, 3)  # (once, not per run: 4096 runs at config 4)
    runs = [_Run(x0, clip_low, clip_upp, m) for x0 in starts]
    active = list(range(len(runs)))
    while active:
        waiting = []
        for k in active:
            r = runs[k]
            while True:
                setulb(m, r.x, low, upp, nbd, r.f, r.g, factr, pgtol, r.wa, r.iwa, r.task, r.lsave, r.isave, r.dsave,
                       maxls, r.ln_task)
                if r.task[0] == 3:  # wants f and g at the current x
                    waiting.append(k)
                    break
                if r.task[0] == 1:  # new iteration
                    r.nit += 1
                    if r.nit >= maxiter:
                        r.task[0], r.task[1] = 5, 504
                    elif r.nfev > maxfun:
                        r.task[0], r.task[1] = 5, 502
                    continue
                r.done = True
                break
        if waiting:
            X = np.array([runs[k].x for k in waiting])
            f, G = fun_batch(X)
            for k, fk, gk in zip(waiting, np.asarray(f, dtype=float), np.asarray(G, dtype=float).reshape(len(waiting), n)):
                r = runs[k]
                r.f[...] = fk  # (in place: the run keeps its two arrays)
                r.g[:] = gk
                r.nfev += 1
        active = waiting
    out = []
    for r in runs:
        warnflag = 0 if r.task[0] == 4 else (1 if (r.nfev > maxfun or r.nit >= maxiter) else 2)
        out.append((r.x.copy(), float(r.f), {"grad": r.g.copy(), "funcalls": r.nfev, "nit": r.nit, "warnflag": warnflag}))
    return out
