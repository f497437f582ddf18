import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "refonly: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def g1():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "g1_state_dict.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_basic():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_basic.npz"), allow_pickle=False)
