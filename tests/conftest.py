import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests need a real device: skip (not fail) them where there is none."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible (the HIP path has no CPU fallback)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def hip_lib():
    """The built HIP library; building is the check (hipcc cross-compiles gfx950 without a GPU)."""
    from surya_amd import build, _lib
    build.build(verbose=False)
    return _lib.lib()
