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


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def oracle():
    from oracle import Oracle
    return Oracle()


def golden_planes(g):
    """Planes are stored for the small cases and regenerated from the seed for full-size ones."""
    if "planes" in g:
        return g["planes"]
    from real3dportrait_amd import synth
    seed, N, C, H, W, scale = g["planes_spec"]
    return synth.synth_planes(int(seed), int(N), int(C), int(H), int(W), float(scale))
