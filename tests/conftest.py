import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_DATA = os.path.join(GOLDEN, "ref_test_data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def ref_files():
    return [os.path.join(REF_DATA, f"meta_test_{i}.fa") for i in (1, 2, 3)]


@pytest.fixture(scope="session")
def gpu_ctx():
    """One HIP context for the whole GPU session; kernels run on torch's current stream."""
    import torch
    from metafast_amd import lib as L
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    ctx = L.Context(0, stream=torch.cuda.current_stream())
    yield ctx
    ctx.close()
