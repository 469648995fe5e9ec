"""Host side of the ChangePoint gradient for any number of regions (CPU only): the window-parameter components that
`GpRegressor._mix_window_terms` / `_mix_window_gradient` assemble from the device's row sums equal the reference's own
expression, 1/2 sum Q o (K_c o (A + A^T) + K_{c+1} o (B + B^T)) with A = -df (1 - f)^T, B = df f^T
(/root/reference/inference/gp/covariance.py:561-594) - with the row sums done in NumPy here instead of on the device."""
from types import SimpleNamespace

import numpy as np
import pytest

from inference_amd.gp import covariance as C
from inference_amd.gp.regression import GpRegressor


@pytest.mark.parametrize("nk", [2, 3, 4])
def test_window_gradient_from_row_sums_equals_the_reference_expression(nk):
    rng = np.random.default_rng(nk)
    n = 37
    x = np.sort(rng.uniform(0, 1, n)).reshape(-1, 1)
    cp = C.ChangePoint(kernels=[C.SquaredExponential] * nk)
    cp.pass_spatial_data(x)
    theta = rng.normal(size=cp.n_params)
    for c, slc in enumerate(cp.cp_slc):
        theta[slc] = [(c + 1) / nk, 0.05 + 0.02 * c]
    # any symmetric Q and symmetric sub-kernel matrices: the identity is algebra, not a property of the kernels
    Q = rng.normal(size=(n, n))
    Q = Q + Q.T
    K = []
    for _ in range(nk):
        M = rng.normal(size=(n, n))
        K.append(M + M.T)
    hw, dws = GpRegressor._mix_window_terms(SimpleNamespace(_mix=cp), theta)
    assert hw.shape == (nk, 2, n) and len(dws) == nk - 1
    assert not hw[0, 0].any() and not hw[nk - 1, 1].any()  # the end regions have one neighbour
    hrows = np.array([[(Q * K[m]) @ hw[m, r] for r in range(2)] for m in range(nk)])  # what the device returns
    got = np.zeros(cp.n_params)
    GpRegressor._mix_window_gradient(cp, dws, hrows, got)
    want = np.zeros(cp.n_params)
    for c, slc in enumerate(cp.cp_slc):
        f, dfs = cp.logistic_and_gradient(cp.x_cp, theta[slc])
        for k, df in enumerate(dfs):
            A = -df[:, None] * (1 - f)[None, :]
            B = df[:, None] * f[None, :]
            want[slc.start + k] = 0.5 * np.sum(Q * (K[c] * (A + A.T) + K[c + 1] * (B + B.T)))
    assert np.allclose(got, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
    assert not got[: cp.cp_slc[0].start].any()  # only the window parameters are written
    # two regions: the row-sum weights are the window weights themselves (the case the device's default rows cover)
    if nk == 2:
        assert np.array_equal(hw[0, 1], cp.weights(cp.x_cp, theta)[0]) and np.array_equal(hw[1, 0], cp.weights(cp.x_cp, theta)[1])
