import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')


def pytest_collection_modifyitems(config, items):
    """Off-device (`pytest tests` on a CPU host) the `gpu` tests are skipped instead of erroring one by one.  On a
    GPU box nothing is skipped: a missing HIP library must fail loudly there (the product path has no fallback)."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='needs a ROCm GPU (run with -m gpu through gpurun)')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def lib():
    """The C-ABI library (built in-tree by __graft_entry__.build())."""
    from luminoth_amd import _lib
    return _lib.load()
