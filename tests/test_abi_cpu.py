"""CPU-only checks of the drop-in boundary: the C-ABI library is built, loads, and exports
every symbol that include/gpmi.h declares; the ctypes table mirrors the header.  No compute
call is made (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HEADER = os.path.join(ROOT, "include", "gpmi.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpmi_[a-z_A-Z0-9]+)\s*\(", text)))


def test_header_declares_the_path():
    syms = declared_symbols()
    for must in ("gpmi_set_data", "gpmi_fit", "gpmi_lml", "gpmi_lml_batch", "gpmi_lml_grad", "gpmi_predict"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from inference_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in gpmi.h but not exported by libgpmi.so"
    assert lib.gpmi_version() == 100


def test_ctypes_table_matches_header():
    from inference_amd import _lib

    assert sorted(_lib.SIGNATURES) == declared_symbols()
    _lib.load()


def test_no_silent_cpu_path():
    """Without a GPU a handle cannot be created: the product path fails loudly."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from inference_amd import _lib

    with pytest.raises(_lib.GpmiUnavailable):
        _lib.Handle(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "inference-tools_amd", "inference_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src, f"{f} references the oracle"


def test_host_side_kernel_mirrors_against_reference_vectors(golden):
    """The O(N) host parts of the covariance classes (labels, bounds, diagonal blocks) need no GPU:
    HeteroscedasticNoise against the reference's outputs (tests/golden/het.npz)."""
    import numpy as np

    import workloads as wl
    from inference_amd.gp.covariance import HeteroscedasticNoise, SquaredExponential, device_plan, heteroscedastic_slice

    g = golden("het")
    x, y, _ = wl.synthetic_dataset(77, 96, 1)
    cov = SquaredExponential() + HeteroscedasticNoise()
    cov.pass_spatial_data(x)
    cov.estimate_hyperpar_bounds(y)
    assert [f"{s}" for s in cov.hyperpar_labels] == list(g["err_labels"])[1:]
    assert np.allclose(np.array(cov.bounds, dtype=float), g["err_bounds"][1:], rtol=1e-12)
    assert device_plan(cov) is not None and heteroscedastic_slice(cov) == slice(2, 98)
    th = g["err_thetas"][1][1:]
    het = cov.components[1]
    Kh, grads = het.covariance_and_gradients(th[heteroscedastic_slice(cov)])
    assert np.array_equal(np.diagonal(Kh), np.exp(2 * th[2:])) and np.count_nonzero(Kh) == 96
    assert len(grads) == 96 and grads[5][5, 5] == 2 * Kh[5, 5] and np.count_nonzero(grads[5]) == 1
    assert het(x[:3], x, th[2:]).shape == (3, 96)
