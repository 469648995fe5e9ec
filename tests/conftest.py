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
