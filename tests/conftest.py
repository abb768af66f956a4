import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def cases():
    import json

    import numpy as np

    data = np.load(os.path.join(GOLDEN, "embbag_cases.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "embbag_cases.json")))["cases"]
    return data, meta


@pytest.fixture(scope="session")
def coracle():
    from oracle.embbag_oracle import COracle

    return COracle()
