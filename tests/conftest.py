import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the reference build in oracle/_ref (build container)")


@pytest.fixture(scope="session")
def refh():
    import _refh
    if not _refh.available():
        pytest.skip("oracle/_ref not built (no /root/reference on this host)")
    return _refh
