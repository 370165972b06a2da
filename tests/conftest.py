import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "go-blosc_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def _have_gpu():
    try:
        import hipblosc
        return hipblosc.lib().hb_init() == 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def hb():
    """The product binding.  GPU tests fail (not skip) when the HIP library is missing or sees no device."""
    import hipblosc
    rc = hipblosc.lib().hb_init()
    assert rc == 0, "libhipblosc.so loaded but no HIP device is usable — GPU tests need the real MI355X"
    return hipblosc


@pytest.fixture(scope="session")
def O():
    import oracle
    oracle.build()
    return oracle
