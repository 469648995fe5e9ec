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


def test_library_exports_nothing_but_the_header():
    """The converse: the dynamic symbol table of the plugin boundary is include/gpmi.h and nothing else (csrc/gpmi.map) -
    no mangled C++ internals, no std:: instantiations, no HIP bookkeeping symbols."""
    import subprocess

    from inference_amd import _lib

    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    defined = sorted({line.split()[-1].split("@")[0] for line in out.splitlines() if line.strip()})
    assert defined == declared_symbols(), sorted(set(defined) ^ set(declared_symbols()))


def test_product_names_no_vendor_solver():
    """rocSOLVER / rocBLAS / hipBLAS(Lt) / hipSOLVER are a yardstick for tools/vendor_yardstick.py only: nothing of the
    product (sources, build recipe, the built library's dependencies) names them."""
    import subprocess

    from inference_amd import _lib

    banned = re.compile(r"rocsolver|rocblas|hipblas|hipsolver|rocsparse|cublas|cusolver", re.I)
    pkg = os.path.join(ROOT, "inference-tools_amd")
    for dirpath, dirs, files in os.walk(pkg):
        dirs[:] = [d for d in dirs if d not in ("build", "__pycache__", ".pytest_cache", "lib")]
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".map")) or f == "Makefile":
                assert not banned.search(open(os.path.join(dirpath, f)).read()), f"{f} names a vendor BLAS / solver"
    assert not banned.search(open(HEADER).read())
    needed = subprocess.run(["readelf", "-d", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert not banned.search(needed), needed


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


def test_change_point_host_side_against_reference_vectors(golden):
    """ChangePoint's O(N) host parts (labels, bounds, the per-point weights the device entry points take)."""
    import numpy as np

    from inference_amd.gp.covariance import ChangePoint, RationalQuadratic, SquaredExponential, WhiteNoise, device_plan

    g = golden("cp")
    x, y = g["x"], g["y"]
    cov = ChangePoint(kernels=[SquaredExponential, RationalQuadratic]) + WhiteNoise()
    cov.pass_spatial_data(x)
    cov.estimate_hyperpar_bounds(y)
    plan = device_plan(cov)
    assert plan is not None and plan[0] == -1 and plan[2] == slice(0, 9) and plan[3] == 9
    cp = ChangePoint(kernels=[SquaredExponential, RationalQuadratic])
    cp.pass_spatial_data(x)
    cp.estimate_hyperpar_bounds(y)
    assert cp.hyperpar_labels == list(g["serq_labels"])[1:]
    assert np.allclose(np.array(cp.bounds, dtype=float), g["serq_bounds"][1:], rtol=1e-12)
    th = g["serq_thetas"][1][1:]
    w = cp.weights(x[:, 0], th)
    f = 1.0 / (1.0 + np.exp(-(x[:, 0] - th[7]) / th[8]))
    assert w.shape == (2, 200) and np.allclose(w[0], 1 - f) and np.allclose(w[1], f)
    kernels, thetas = cp.device_terms(th)
    assert kernels == [0, 1] and [t.size for t in thetas] == [3, 4]
    three = ChangePoint(kernels=[SquaredExponential] * 3)
    three.pass_spatial_data(x)
    w3 = three.weights(x[:, 0], np.array([0, 0, 0, 0, 0, 0, 0, 0, 0, 0.3, 0.05, 0.7, 0.05]))
    f0 = 1.0 / (1.0 + np.exp(-(x[:, 0] - 0.3) / 0.05))
    f1 = 1.0 / (1.0 + np.exp(-(x[:, 0] - 0.7) / 0.05))
    # the coefficient recursion of covariance.py:529-544: [1 - f0, f0 (1 - f1), f1]
    assert np.allclose(w3[0], 1 - f0) and np.allclose(w3[1], f0 * (1 - f1)) and np.allclose(w3[2], f1)
