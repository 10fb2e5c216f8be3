import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    import __graft_entry__ as g

    g.build()
    return True


@pytest.fixture(scope="session")
def ctx(built):
    import torch

    import msm_webgpu_amd as m

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    c = m.MsmContext(0)
    yield c
    c.close()
