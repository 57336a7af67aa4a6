import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def vectors():
    with open(os.path.join(ROOT, "tests", "golden", "plonk_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def ctx():
    """One library context on GPU 0.  No fallback: a missing library or GPU is a failure."""
    import plonk_amd
    c = plonk_amd.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def prover():
    from plonk_amd.device import DeviceProver
    return DeviceProver(0)
