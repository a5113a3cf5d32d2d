import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def golden_dbs(tmp_path_factory):
    """Extracts the reference's 10 fixture .kreeq databases (tests/golden/kreeq_dbs.tar.gz)."""
    import tarfile

    d = tmp_path_factory.mktemp("kreeq_dbs")
    with tarfile.open(os.path.join(ROOT, "tests", "golden", "kreeq_dbs.tar.gz")) as tar:
        tar.extractall(d)
    return str(d)
