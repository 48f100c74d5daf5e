import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import loader
    loader.build()
    return loader.load()


@pytest.fixture(params=["default", "lane"])
def both_passes(request, monkeypatch):
    """Run a GPU test twice: with the library's own choice of streaming pass and with the
    lane-per-rollout pass forced wherever it applies (SMPC_PASS is read at smpc_create)."""
    if request.param == "lane":
        monkeypatch.setenv("SMPC_PASS", "lane")
    else:
        monkeypatch.delenv("SMPC_PASS", raising=False)
    return request.param
