"""The drop-in boundary is plain C: a C99 client (tests/c_abi/fit_predict.c) that uses nothing but include/gpmi.h
must compile and link against libgpmi.so with gcc (CPU suite) and run correctly on the GPU box (-m gpu)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "inference-tools_amd", "inference_amd", "lib")
SRC = os.path.join(ROOT, "tests", "c_abi", "fit_predict.c")
SRC_DENSE = os.path.join(ROOT, "tests", "c_abi", "dense_append.c")


def _build(tmp_path, src=SRC):
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    if not os.path.exists(os.path.join(LIBDIR, "libgpmi.so")):
        pytest.skip("libgpmi.so not built")
    exe = os.path.join(str(tmp_path), os.path.splitext(os.path.basename(src))[0])
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"), src, "-o", exe,
           "-L" + LIBDIR, "-lgpmi", "-lm", "-Wl,-rpath," + LIBDIR]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_header_is_plain_c_and_links(tmp_path):
    """gcc -std=c99 -pedantic -Werror accepts the header; every symbol the client uses resolves at link time."""
    exe = _build(tmp_path)
    assert os.path.exists(exe)
    assert os.path.exists(_build(tmp_path, SRC_DENSE))


@pytest.mark.gpu
def test_c_client_fits_and_predicts(tmp_path):
    exe = _build(tmp_path)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.startswith("ok ")


@pytest.mark.gpu
def test_c_client_dense_kernel_and_append(tmp_path):
    """The dense entry points (a Matern-3/2 covariance built by the C program itself) and gpmi_append_point, from C:
    residual, inverse, predict pieces and the appended model all check out against host arithmetic in the client."""
    exe = _build(tmp_path, SRC_DENSE)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.startswith("ok ")
