import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = ("scene_small_noise", "scene_small_smooth", "scene_capped")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def scene_inputs(g):
    mr = int(g["max_resolution"])
    return list(g["imgs"]), g["rots"], g["intrs"], (1400 if mr < 0 else mr)


def n_patches(g, prefix="mb"):
    n = 0
    while f"{prefix}_irange_{n}" in g:
        n += 1
    return n


@pytest.fixture(scope="session")
def oracle():
    import pano_oracle
    pano_oracle.build()
    return pano_oracle


@pytest.fixture(scope="session")
def eng():
    """The HIP engine; GPU tests fail loudly (not skip) if it cannot start."""
    from pano360_amd import engine
    return engine.engine()
