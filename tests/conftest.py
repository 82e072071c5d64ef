import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Libraries are prebuilt and travel with the snapshot; (re)build here only if missing/stale."""
    import __graft_entry__ as g
    g.build()


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    return oracle_py.Oracle()


@pytest.fixture(scope="session")
def glref():
    import oracle_py
    if not oracle_py.glref_available():
        pytest.skip("reference checkout / llvmpipe harness not present (only in the build container)")
    try:
        return oracle_py.GLRef.get()
    except Exception as e:  # llvmpipe missing
        pytest.skip(f"llvmpipe unavailable: {e}")
