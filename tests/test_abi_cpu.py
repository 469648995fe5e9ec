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
