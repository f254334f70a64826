import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "part-based-3d-reconstruction_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    # stay inside the CPU share of the box (16 per GPU on the GPU box; 8 here): OpenMP's default is every
    # hardware thread it can see, which oversubscribes badly on small inputs
    orc.set_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    return orc


@pytest.fixture(scope="session")
def pb3d_gpu():
    """The product package with a live device context; GPU tests fail loudly if the HIP
    extension is missing or no device is visible (there is no CPU fallback to hide behind)."""
    import pb3d
    if not os.path.exists(pb3d._lib.LIB_PATH):      # fresh checkout: compile the HIP extension in tree (hipcc, gfx950)
        import __graft_entry__
        __graft_entry__.build()
    assert os.path.exists(pb3d._lib.LIB_PATH), "libpb3d.so not built"
    assert pb3d._lib.device_count() >= 1, "no MI355X visible"
    pb3d._lib.ctx()
    return pb3d
