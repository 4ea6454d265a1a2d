import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build whatever is missing or older than its sources (every builder is mtime-incremental, so a
    prebuilt, up-to-date library that travelled to the GPU box is left alone)."""
    from talc_amd import build as B
    B.build_synth()
    B.build_pure()
    B.build_oracle()
    B.build_hip()
    B.build_cli()
    yield


def has_gpu():
    try:
        from talc_amd import lib as T
        return T.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_pair():
    """A small oracle/product table pair uploaded to GPU 0 (session-wide)."""
    import parity_util as PU
    pair = PU.Pair(target_kmers=400_000, k=21, seed=11)
    pair.upload(0)
    return pair
