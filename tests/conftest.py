import os
import sys

import pytest

# the oracle's OpenMP loops are tiny at test sizes: a 128-thread team costs more than it saves
os.environ.setdefault("OMP_NUM_THREADS", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle as orc
    orc.build()
    return orc
