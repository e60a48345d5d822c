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


def pytest_sessionstart(session):
    """The built artefacts are git-ignored: on a fresh checkout build libvgh.so (hipcc cross-compiles without a GPU) so the ABI
    tests have something to load.  A machine without hipcc keeps whatever is there -- the tests then fail loudly, by design."""
    import shutil

    lib = os.path.join(ROOT, "head_detector_amd", "libvgh.so")
    if not os.path.exists(lib) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        from head_detector_amd import build as b

        b.build_lib(force=False, verbose=False)


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
