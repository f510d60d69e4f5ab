import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def ref():
    """The compiled reference (oracle/_ref), or None where it has not been built/shipped."""
    import oracle_lib
    return oracle_lib.load_ref() if oracle_lib.have_ref() else None


@pytest.fixture(scope="session")
def forced_vectors():
    return np.load(os.path.join(GOLDEN, "forced_vectors.npz"))


@pytest.fixture(scope="session")
def clip_vectors():
    return np.load(os.path.join(GOLDEN, "clip.npz"))


@pytest.fixture(scope="session")
def golden_json():
    def load(name):
        with open(os.path.join(GOLDEN, name)) as f:
            return json.load(f)
    return load


@pytest.fixture(scope="session")
def hiplib():
    """libdetexhip.so through the reference-API ctypes binding (host-pointer tier)."""
    import oracle_lib
    from detex_amd import binding
    binding.load()
    return oracle_lib.DetexAPI(binding.LIB_PATH)
