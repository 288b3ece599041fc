import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure); compiled on demand with gcc."""
    import oracle as _oracle
    _oracle.build()
    _oracle.lib()
    return _oracle


@pytest.fixture(scope="session")
def dr():
    """The product API on cuda:0; fails loudly when the GPU or the HIP library is missing."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test without a GPU"
    from nvdiffrast_amd import _capi
    _capi.load()
    import nvdiffrast_amd.torch as dr_
    return dr_
