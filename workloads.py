"""
Synthetic workloads for the GP hot path.

Shared by bench.py, the tests and tests/golden/make_golden.py so that the HIP
path, the CPU oracle and the imported reference all consume identical arrays.
Arrays are regenerated from seeds (PCG64 streams are stable across NumPy
versions); fixtures therefore hold only outputs.
"""
import numpy as np

SE = 0
RQ = 1


def synthetic_dataset(cfg: int, n: int, d: int):
    """x ~ U(0,1)^(n,d); y = sin(2 pi (x.w)/sqrt(d)) + 0.5 cos(3 x0) + 0.1 N(0,1);
    y_err = 0.1 (keeps cond(K) small enough that 1e-10 parity is attainable)."""
    rng = np.random.default_rng(20250614 + cfg)
    x = rng.uniform(0, 1, (n, d))
    w = rng.normal(size=d)
    y = (
        np.sin(2 * np.pi * (x @ w) / np.sqrt(d))
        + 0.5 * np.cos(3 * x[:, 0])
        + 0.1 * rng.normal(size=n)
    )
    y_err = np.full(n, 0.1)
    return x, y, y_err


def query_points(cfg: int, m: int, d: int):
    """Prediction / candidate points: U(0,1)^(m,d), own stream."""
    rng = np.random.default_rng(77000 + cfg)
    return rng.uniform(0, 1, (m, d))


def timing_theta(kernel: int, y, d: int):
    """Fixed hyper-parameters for timing runs: theta = [mean | cov] with
    theta_mean = mean(y), ln a = ln std(y), ln l_k = ln 0.5, RQ: ln kappa = 0."""
    cov = [np.log(y.std())] + ([0.0] if kernel == RQ else []) + [np.log(0.5)] * d
    return np.array([y.mean()] + cov)


def theta_set(kernel: int, y, d: int, count: int, seed: int = 5):
    """`count` hyper-parameter vectors around timing_theta (first one is
    timing_theta itself), spread over amplitude and length-scales."""
    base = timing_theta(kernel, y, d)
    rng = np.random.default_rng(seed)
    out = [base]
    for _ in range(count - 1):
        t = base.copy()
        t[0] += 0.1 * rng.normal()
        t[1] += rng.uniform(-1.0, 1.0)
        if kernel == RQ:
            t[2] = rng.uniform(-1.0, 2.0)
            t[3:] += rng.uniform(-0.7, 0.9, size=d)
        else:
            t[2:] += rng.uniform(-0.7, 0.9, size=d)
        out.append(t)
    return np.array(out)


def theta_grid_cfg3(y, d: int):
    """Config 3: 64 = 8 x 8 grid over ln a in [ln std - 1, ln std + 1] and a
    common ln l in [ln 0.2, ln 2], ln kappa = 0, RQ kernel."""
    s = np.log(y.std())
    out = []
    for la in np.linspace(s - 1, s + 1, 8):
        for ll in np.linspace(np.log(0.2), np.log(2.0), 8):
            out.append([y.mean(), la, 0.0] + [ll] * d)
    return np.array(out)
