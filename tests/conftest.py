import glob
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# The library honours its DQ_* overrides (forced code paths, fault injection) only under DQ_DEBUG_FLAGS=1
# (dq_runtime.h: env()); the tests force paths all the time.  (bench.py runs without it: production behaviour.)
os.environ["DQ_DEBUG_FLAGS"] = "1"

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
ASSET_DIR = os.path.join(GOLDEN_DIR, "assets")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        return json.load(f)


def asset_names():
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(ASSET_DIR, "*")))


def load_asset(name: str) -> np.ndarray:
    return np.fromfile(os.path.join(ASSET_DIR, name), dtype=np.uint8)


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def backend_lib():
    """The product's shared library, built if needed (hipcc cross-compiles without a GPU)."""
    from deltaq_amd import build as dq_build
    dq_build.build()
    from deltaq_amd import _abi
    return _abi.load()
