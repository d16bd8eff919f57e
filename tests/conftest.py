import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "scann--material_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The in-tree C-ABI library; building it is __graft_entry__.build()'s job."""
    from scann import _hip

    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    return _hip.load_library()
