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


def pytest_sessionfinish(session, exitstatus):
    """Write what the parity checks observed (tests/parity_log.py) to profiles/r06_parity_observed.json — on a GPU box
    only, merged over the file's previous content."""
    import json
    try:
        import torch
        from parity_log import OBSERVED
    except Exception:
        return
    if not OBSERVED or not torch.cuda.is_available():
        return
    path = os.path.join(ROOT, 'profiles', 'r06_parity_observed.json')
    old = {}
    if os.path.exists(path):
        try:
            old = json.load(open(path)).get('observed', {})
        except Exception:
            old = {}
    old.update(OBSERVED)
    doc = {'what': 'largest errors the `-m gpu` parity tests observed (tests/parity_log.py): max = max |got - reference| '
                   '(or the quantity named by the key), tolerance_used = max over elements of error / (atol + rtol |ref|); '
                   'keys name the test and the compared quantity',
           'device': torch.cuda.get_device_name(0), 'observed': dict(sorted(old.items()))}
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'w') as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    if os.path.isdir(out):          # gpurun merges gpurun_out/ back into the build container
        with open(os.path.join(out, 'r06_parity_observed.json'), 'w') as f:
            json.dump(doc, f, indent=1, sort_keys=True)
