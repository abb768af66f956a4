import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain ``pytest tests/`` on a box without a ROCm device skips the gpu-marked tests instead of failing 125 of them.
    When GPU tests were ASKED for (``-m gpu``, as the round-end driver does, or PARAM_AMD_REQUIRE_GPU=1) nothing is
    skipped: a missing device or library then fails loudly."""
    if "gpu" in (config.getoption("-m") or "") or os.environ.get("PARAM_AMD_REQUIRE_GPU") == "1":
        return
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no ROCm device (run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


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
