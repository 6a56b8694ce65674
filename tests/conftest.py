import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fixture_json():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "boojum_fixture.json")) as f:
        return json.load(f)


def pytest_collection_finish(session):
    """PyTorch-ROCm ships its own HIP runtime: the -m gpu tests must get it loaded BEFORE libboojum_hip.so pulls the system copy in
    (tests/gpu_util.py:ctx does that at first use).  A test module that loads the library while it is being imported breaks
    every GPU test after it with "no usable HIP device" — refuse to start instead."""
    try:
        maps = open("/proc/self/maps").read()
    except OSError:
        return
    if "libboojum_hip" in maps:
        pytest.exit("a test module loaded libboojum_hip.so at import time (module-level load_library(), or an object whose "
                    "__getattr__ loads it: pytest probes attributes while collecting); load it inside the tests", returncode=3)
