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


def golden(name: str):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def flame_model():
    """Synthetic FLAME constants (seed 3) on the reference's real v_template (tests/golden/flame_decode.npz)."""
    from oracle import flame_oracle as fo

    g = golden("flame_decode.npz")
    return fo.synthetic_flame_model(seed=int(g["seed"]), v_template=g["v_template"].astype(np.float64))


@pytest.fixture(scope="session")
def gpu_lib():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from head_detector_amd import _lib

    return _lib.load()  # raises loudly if libvgh.so is missing: GPU tests must never run on a fallback
