import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; on a machine without the amdgpu device node they are skipped,
    # never silently passed.  On a GPU box a missing libmirt.so makes them fail.
    if os.path.exists("/dev/kfd"):
        return
    skip = pytest.mark.skip(reason="no GPU device node (/dev/kfd)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def rta():
    import raytracing_amd
    return raytracing_amd
