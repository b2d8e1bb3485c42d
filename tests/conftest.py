import os
import sys
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(__file__))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: larger CPU cases")


def _has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def has_gpu():
    return _has_gpu()
