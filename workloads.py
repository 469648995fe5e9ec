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


# ---------------------------------------------------------------------------------------------
# linear-inversion problems (GpLinearInverter): model matrix, data, parameter positions
# ---------------------------------------------------------------------------------------------
def linv_problem(name):
    """`deconv`: 1-D deconvolution in the spirit of the reference's own test (32 data values, 64 model
    parameters on [-1, 1], Gaussian blur; tests/gp/test_GpLinearInverter.py:44-67), written out
    independently: three peaks, pixel-integrated Gaussian response.
    `tomo`: 2-D, 400 parameters on a 20 x 20 grid seen through 300 random smooth 'chords' (each row of A a
    normalised Gaussian ridge across the grid) - sizes that exercise the padded tiles (300 -> 384, 400 -> 512).
    Returns (positions (n, d), A (m, n), y (m,), y_err (m,))."""
    from scipy.special import erf

    if name == "deconv":
        m, n = 32, 64
        x = np.linspace(-1.0, 1.0, n)
        t = np.linspace(-1.0, 1.0, m)
        half = 0.5 * (x[1] - x[0])
        width = 0.075
        truth = 1.0 / (1 + (x / 0.1) ** 2) + 0.8 / (1 + ((x - 0.3) / 0.15) ** 2) + 0.3 / (1 + ((x + 0.45) / 0.1) ** 2)
        cdf = lambda z: 0.5 * (1.0 + erf(z / (np.sqrt(2.0) * width)))  # noqa: E731
        A = cdf(t[:, None] + half - x[None, :]) - cdf(t[:, None] - half - x[None, :])
        rng = np.random.default_rng(20250614 + 101)
        y_err = np.full(m, 0.02)
        y = A @ truth + rng.normal(size=m) * y_err
        return x.reshape(n, 1), A, y, y_err
    if name == "tomo":
        g = 20
        m, n = 300, g * g
        ax = (np.arange(g) + 0.5) / g
        X, Y = np.meshgrid(ax, ax, indexing="ij")
        pos = np.stack([X.ravel(), Y.ravel()], axis=1)
        rng = np.random.default_rng(20250614 + 102)
        ang = rng.uniform(0, np.pi, m)
        off = rng.uniform(-0.35, 0.35, m)
        # distance of every pixel centre to chord i (through the centre + offset along the normal)
        dist = (pos[None, :, 0] - 0.5) * np.cos(ang)[:, None] + (pos[None, :, 1] - 0.5) * np.sin(ang)[:, None] - off[:, None]
        A = np.exp(-0.5 * (dist / 0.04) ** 2)
        A /= A.sum(axis=1, keepdims=True)
        truth = np.exp(-((pos[:, 0] - 0.4) ** 2 + (pos[:, 1] - 0.55) ** 2) / 0.03) + 0.5 * np.exp(
            -((pos[:, 0] - 0.7) ** 2 + (pos[:, 1] - 0.3) ** 2) / 0.01
        )
        y_err = np.full(m, 0.01)
        y = A @ truth + rng.normal(size=m) * y_err
        return pos, A, y, y_err
    raise ValueError(name)


def linv_thetas(name, kid, white_noise=False):
    """Three hyper-parameter vectors [mean, ln a, (ln kappa), ln l.., (ln sigma_n)] per problem."""
    d = 1 if name == "deconv" else 2
    base_l = np.log(0.15 if name == "deconv" else 0.2)
    out = []
    for i, (mean, lna, dl) in enumerate([(0.1, 0.0, 0.0), (0.3, -0.5, 0.4), (0.0, 0.4, -0.3)]):
        th = [mean, lna]
        if kid == RQ:
            th.append(0.3 - 0.2 * i)
        th += [base_l + dl + 0.1 * k for k in range(d)]
        if white_noise:
            th.append(np.log(0.05) + 0.3 * i)
        out.append(np.array(th))
    return out
