import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import rtlibs  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    z = np.load(os.path.join(rtlibs.GOLDEN_DIR, "reference_vectors.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_meta():
    with open(os.path.join(rtlibs.GOLDEN_DIR, "reference_meta.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    return rtlibs.Oracle()


@pytest.fixture(scope="session")
def scene_paths():
    return [os.path.join(rtlibs.DATA_DIR, f"scene_{i}.txt") for i in range(3)]
