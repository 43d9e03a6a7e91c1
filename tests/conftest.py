import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) -- run with -m gpu")


@pytest.fixture(scope="session")
def oracle():
    from oracle.bindings import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle.bindings import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref not built and /root/reference absent")
    return Ref()


@pytest.fixture(scope="session")
def sx():
    """The product package with its C-ABI library built (hipcc cross-compiles without a GPU)."""
    from sextans_amd import build as b
    b.build()
    import sextans_amd
    sextans_amd.api.lib()
    return sextans_amd


@pytest.fixture(scope="session")
def engine(sx):
    if sx.device_count() < 1:
        pytest.fail("no gfx950 device visible: GPU tests cannot run (there is no CPU fallback)")
    e = sx.Engine(0)
    yield e
    e.close()
