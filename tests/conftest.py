import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    import __graft_entry__ as g
    g.build()
    return g


@pytest.fixture(scope="session")
def pkg(built):
    return built.load_package()


@pytest.fixture(scope="session")
def oracle(built):
    import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def synth(pkg):
    from tamcmc_c_amd import synth as s
    return s
