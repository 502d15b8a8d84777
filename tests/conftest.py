import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


@pytest.fixture(scope="session")
def hooks_lib():
    """libstenos with the test switches compiled in (tests/hooks): only the tests that need a switch use it"""
    from _libs import load_hooks_library

    return load_hooks_library()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from _libs import load_oracle

    return load_oracle()


@pytest.fixture(scope="session")
def ref_det():
    from _libs import load_ref

    lib = load_ref(det=True)
    if lib is None:
        pytest.skip("oracle/_ref not built (only available in the build container)")
    return lib
