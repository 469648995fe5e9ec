import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "inference-tools_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_report_header(config):
    """Which native library the suite runs against: GPMI_LIB lets an environment variable put another build of libgpmi.so
    under the tests (A/B timing of kernel variants) - the header says which file was loaded, and its version."""
    try:
        from inference_amd import _lib

        path = _lib.LIB_PATH
        if not os.path.exists(path):
            return f"libgpmi: {path} (NOT BUILT)"
        return (f"libgpmi: {path} (gpmi_version {_lib.load().gpmi_version()}"
                f"{', selected by GPMI_LIB' if os.environ.get('GPMI_LIB') else ''})")
    except Exception as err:  # the header must never break a run
        return f"libgpmi: not loadable here ({type(err).__name__}: {err})"


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            path = os.path.join(GOLDEN, f"{name}.npz")
            if not os.path.exists(path):
                pytest.skip(f"golden fixture {name}.npz not generated")
            cache[name] = dict(np.load(path, allow_pickle=False))
        return cache[name]

    return load


def pytest_sessionfinish(session, exitstatus):
    """Destroy every live device context while the interpreter and the HIP runtime are fully alive (the library also does
    this from atexit: a context torn down later, during interpreter shutdown, can end in the runtime's static destructors)."""
    try:
        mod = sys.modules.get("inference_amd._lib")
        if mod is not None:
            mod._close_all_handles()
    except Exception:
        pass


def pytest_terminal_summary(terminalreporter):
    """Achieved errors of the GPU parity comparisons (tests/test_gpu_parity.py logs every `check`): the worst
    ratio error / tolerance per test and label."""
    mod = sys.modules.get("test_gpu_parity") or sys.modules.get("tests.test_gpu_parity")
    rows = getattr(mod, "ACHIEVED", None) if mod else None
    if not rows:
        return
    worst = {}
    for test, what, r, tol in rows:
        key = (test, what)
        if key not in worst or r > worst[key][0]:
            worst[key] = (r, tol)
    tr = terminalreporter
    tr.section("achieved parity errors (max per test / quantity)")
    # worst LAST, 30 lines: the driver's record keeps the tail of the output, so the entries closest to their
    # tolerance are the ones that survive truncation
    ranked = sorted(worst.items(), key=lambda kv: (kv[1][0] / kv[1][1] if kv[1][1] > 0 else 0))
    tr.write_line(f"{len(worst)} quantities, {len(rows)} comparisons in all; the 30 closest to their tolerance, worst last:")
    for (test, what), (r, tol) in ranked[-30:]:
        tr.write_line(f"{r:9.2e}  (tol {tol:7.1e})  {test} :: {what}")
    # GPMI_PARITY_TABLE=<file>: the whole table (profiles/rNN_parity_errors.txt is such a file)
    path = os.environ.get("GPMI_PARITY_TABLE")
    if path:
        with open(path, "w") as f:
            f.write(f"# achieved parity errors of one `pytest -m gpu` run: {len(worst)} quantities, {len(rows)} comparisons; "
                    "max per test / quantity, closest to the tolerance last\n")
            for (test, what), (r, tol) in ranked:
                f.write(f"{r:9.2e}  (tol {tol:7.1e})  {test} :: {what}\n")
