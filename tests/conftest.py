import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: long-running oracle case (skipped unless TEASER_SLOW=1)")


def pytest_collection_modifyitems(config, items):
    # the certifier tests go LAST: the warm-up thread started below (rocBLAS / rocSOLVER code objects, ~110 s of
    # host-side loading) then runs behind the rest of the GPU suite instead of in front of it
    items.sort(key=lambda it: 1 if "certifier" in it.nodeid and "gpu" in it.keywords else 0)
    if any("test_gpu_certifier" in it.nodeid and "gpu" in it.keywords for it in items) and \
            "not gpu" not in (config.getoption("-m") or ""):
        try:
            import importlib
            tp = importlib.import_module("teaser-plusplus_amd")
            if tp.device_count() > 0:
                tp.certifier_warmup(0)
        except Exception:  # (no library / no device: the tests themselves report it)
            pass
    if os.environ.get("TEASER_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="set TEASER_SLOW=1")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)
