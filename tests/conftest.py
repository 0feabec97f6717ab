import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_sessionstart(session):
    # the GPU boxes show every core of the host but grant a 16-core share: without a cap the oracle's CPU convolutions run
    # oversubscribed (measured: the fp64 C2 oracle step 7 s on 8 threads here, minutes there)
    import torch
    torch.set_num_threads(min(os.cpu_count() or 1, 16))


@pytest.fixture(scope="session")
def lib():
    """Build (if needed) and load libnvae_hip.so."""
    from nvae_tf_amd import build, _lib
    build.build(verbose=False)
    return _lib.load()


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
