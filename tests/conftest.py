import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box)")


@pytest.fixture(scope="session")
def earth():
    import raytracing_amd as rt
    return rt.earthmap()


@pytest.fixture(scope="session", autouse=True)
def _test_libs():
    import oracle_lib
    oracle_lib.build_test_libs()
