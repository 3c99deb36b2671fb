import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle_built():
    """Compile the C oracle (and the reference when its sources are present)."""
    from oracle import pyoracle as po
    po.build(ref=os.path.isdir("/root/reference/src"))
    return po
