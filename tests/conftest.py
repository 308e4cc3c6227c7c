import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _fresh_library():
    """Rebuild libfyc_hip.so when a source changed (no-op when the object digests match)."""
    import shutil
    from followyourclick_amd import _build
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        _build.build(verbose=False)
    yield
