import os
import sys

import pytest

# Tests switch the library's kernel-selection knobs inside one process: the library caches every knob at first use
# unless this is set before its first call.
os.environ.setdefault('STMI_ENV_LIVE', '1')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'style-transfer_amd'), os.path.join(ROOT, 'style-transfer_amd', 'src')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a gfx950 GPU and the built libstmi355.so')


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('this test is marked gpu and needs a visible MI355X')
    return torch.device('cuda:0')
