import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_artifacts():
    """Build products are git-ignored and normally arrive prebuilt; on a tree without them (fresh checkout on a box that has
    hipcc / gcc) build them once.  No-ops when they are up to date."""
    from orbslam2_nmi_amd import build as nmi_build
    nmi_build.build()
    from oracle import binding as oracle_binding
    oracle_binding.build()


@pytest.fixture(scope="session")
def golden_pairs():
    return np.load(os.path.join(GOLDEN, "pairs_64x48.npz"))


@pytest.fixture(scope="session")
def golden_grid():
    return np.load(os.path.join(GOLDEN, "grid_64x48.npz"))


@pytest.fixture(scope="session")
def golden_kat():
    return np.load(os.path.join(GOLDEN, "kat_640x480.npz"))


def dense_joint(g, tag):
    j = np.zeros(65536, np.uint32)
    j[g[f"{tag}/joint_idx"]] = g[f"{tag}/joint_cnt"]
    return j.reshape(256, 256)
