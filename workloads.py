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


def cfg5_ladder(gp, k: int, n_temps: int = 8, t_max: float = 100.0):
    """Ladder k of BASELINE config 5: `n_temps` GibbsChains at temperatures
    10**linspace(0, log10(t_max), n_temps) over the hyper-parameters of `gp` (flat box prior = hp_bounds), posterior
    = gp.marginal_likelihood, every random generator seeded from k so that a run does not depend on how the ladders
    are grouped or sharded."""
    import random

    from numpy.random import default_rng

    from inference_amd.mcmc import GibbsChain, ParallelTempering

    start = np.array([0.5 * (a + b) for a, b in gp.hp_bounds])
    start[0] = gp.y.mean()
    widths = np.array([0.05 * (b - a) for a, b in gp.hp_bounds])
    # every chain of every ladder starts at the same point, and a GibbsChain evaluates its start twice when it is built
    # (validation + first log-probability, as the reference's does): that likelihood is computed once per model here,
    # not 2 x 8 x n_ladders times one by one; the chains then get the model's own bound method back, which is what lets
    # the tempering driver find the batched form
    memo = gp.__dict__.setdefault("_cfg5_start_memo", {})

    def start_posterior(theta):
        key = np.asarray(theta, dtype=float).tobytes()
        if key not in memo:
            memo[key] = gp.marginal_likelihood(theta)
        return memo[key]

    chains = []
    for t_i, temp in enumerate(10.0 ** np.linspace(0.0, np.log10(t_max), n_temps)):
        ch = GibbsChain(posterior=start_posterior, start=start, widths=widths, temperature=float(temp),
                        display_progress=False)
        ch.posterior = gp.marginal_likelihood
        for i, b in enumerate(gp.hp_bounds):
            ch.set_boundaries(i, b)
        ch.rng = default_rng(100_000 * k + 100 * t_i)
        for i, par in enumerate(ch.params):
            par.rng = default_rng(100_000 * k + 100 * t_i + 1 + i)
        chains.append(ch)
    pt = ParallelTempering(chains)
    pt.rng = default_rng(7_000_000 + k)
    pt.pair_choice = random.Random(9_000_000 + k).choice
    return pt


class Matern32Math:
    """A covariance function that is NOT one of the library's device kernels: Matern-3/2 with one length-scale per
    dimension, K(u, v) = a^2 (1 + sqrt3 r) exp(-sqrt3 r), r^2 = sum_k ((u_k - v_k) / l_k)^2, theta = [ln a, ln l_1..d],
    written against the plugin contract only (covariance.py:8-44).  Mixed with the reference's CovarianceFunction it
    generates tests/golden/plugin.npz, mixed with this package's it exercises the dense device entry points."""

    def __init__(self, hyperpar_bounds=None):
        self.bounds = hyperpar_bounds

    def pass_spatial_data(self, x):
        self.x = np.asarray(x, dtype=float)
        d = self.x.shape[1]
        self.n_params = d + 1
        self.hyperpar_labels = ["Matern32 log-amplitude"] + [f"Matern32 log-scale {i}" for i in range(d)]
        self.dx2 = (self.x[:, None, :] - self.x[None, :, :]) ** 2

    def estimate_hyperpar_bounds(self, y):
        s = np.log(np.std(y))
        span = np.ptp(self.x, axis=0)
        self.bounds = [(s - 4, s + 4)] + [(np.log(w) - 4, np.log(w) + 2) for w in span]

    @staticmethod
    def _k_of_r(a2, r):
        return a2 * (1 + np.sqrt(3.0) * r) * np.exp(-np.sqrt(3.0) * r)

    def __call__(self, u, v, theta):
        a2, l = np.exp(2 * theta[0]), np.exp(theta[1:])
        r = np.sqrt((((u[:, None, :] - v[None, :, :]) / l[None, None, :]) ** 2).sum(axis=2))
        return self._k_of_r(a2, r)

    def build_covariance(self, theta):
        a2, l = np.exp(2 * theta[0]), np.exp(theta[1:])
        r = np.sqrt((self.dx2 / l[None, None, :] ** 2).sum(axis=2))
        return self._k_of_r(a2, r) + 1e-10 * a2 * np.eye(self.x.shape[0])

    def covariance_and_gradients(self, theta):
        a2, l = np.exp(2 * theta[0]), np.exp(theta[1:])
        z = self.dx2 / l[None, None, :] ** 2
        r = np.sqrt(z.sum(axis=2))
        K = self._k_of_r(a2, r) + 1e-10 * a2 * np.eye(self.x.shape[0])
        e = 3.0 * a2 * np.exp(-np.sqrt(3.0) * r)
        return K, [2.0 * K] + [z[:, :, k] * e for k in range(l.size)]

    def gradient_terms(self, v, x, theta):
        a2, l = np.exp(2 * theta[0]), np.exp(theta[1:])
        dx = x - v[None, :]
        r = np.sqrt(((dx / l[None, :]) ** 2).sum(axis=1))
        A = 3.0 * dx / (l[None, :] ** 2 * (1 + np.sqrt(3.0) * r)[:, None])
        return A.T, 3.0 * a2 / l**2


def plugin_problem():
    n, d = 80, 2
    rng = np.random.default_rng(8080)
    x = rng.uniform(0, 2, (n, d))
    y = np.sin(2 * x[:, 0]) * np.cos(1.5 * x[:, 1]) + 0.3 * x[:, 0] + 0.05 * rng.normal(size=n)
    e = np.full(n, 0.05)
    pts = rng.uniform(0, 2, (30, d))
    base = np.array([y.mean(), np.log(y.std()), np.log(0.8), np.log(1.1)])
    thetas = np.array([base + 0.15 * rng.standard_normal(4) for _ in range(3)])
    return x, y, e, pts, thetas
