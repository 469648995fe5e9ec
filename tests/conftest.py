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
