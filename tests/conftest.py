import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "course-assignment-danielhalachev_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (its directory name has hyphens, so it is imported by string)."""
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def scenes(pkg):
    return pkg.scenes


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_api
    oracle_api.lib()
    return oracle_api
