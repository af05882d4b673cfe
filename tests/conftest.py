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


def _gpu_present():
    """False only when the library loads and counts zero devices; a missing or broken libmbgc_hip.so is NOT a reason to
    skip (the gpu tests then fail loudly, which is what a GPU box must show)"""
    from mbgc_amd import binding
    try:
        L = binding.lib()
    except Exception:
        return True
    return L.swsem_device_count() > 0


def pytest_collection_modifyitems(config, items):
    """`gpu` tests are skipped, not failed, on a host without a device (plain `pytest tests` in the build container)"""
    if not any("gpu" in it.keywords for it in items) or _gpu_present():
        return
    skip = pytest.mark.skip(reason="no HIP device on this host")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
