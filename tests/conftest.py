import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_ROOT = os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd")
for p in (ROOT, PKG_ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def built_library():
    """The in-tree HIP library and its objects, (re)built when a source is newer (hipcc cross-compiles for gfx950 without a GPU;
    a no-op after ``__graft_entry__.build()``)."""
    from umhsnerf import build

    return build.build_lib(verbose=False)
