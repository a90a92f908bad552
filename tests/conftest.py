import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for _p in (str(ROOT), str(ROOT / "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """the real reference (oracle/_ref) when it has been built; tests that need it skip otherwise"""
    from oracle.pyoracle import RefGgml, ref_available
    if not ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference; run `make -C oracle ref`)")
    return RefGgml("avx2")


GOLDEN = ROOT / "tests" / "golden"
TYPES = ["q4_0", "q8_0", "q4_K", "q5_K", "q6_K", "q4_1", "q5_0", "q5_1", "q2_K", "q3_K", "iq4_nl", "iq4_xs"]


@pytest.fixture(scope="session", params=TYPES)
def golden(request):
    import numpy as np
    return np.load(GOLDEN / f"golden_{request.param}.npz")
