import os
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('no HIP device')
    return torch.device('cuda:0')


@pytest.fixture(scope='session')
def lib():
    from next3d_amd import _lib
    return _lib.lib()
