import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the C-ABI library is a build artefact (git-ignored): build it once if this checkout does not have it yet
    # (hipcc cross-compiles gfx950 without a GPU; on the GPU box the snapshot already carries the built .so)
    lib = os.path.join(ROOT, "lorenzcycletoolkit_amd", "liblec_hip.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
