import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
for p in (os.path.join(ROOT, "dl-dkd_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden"),
          os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no libdldkd_hip.so (built artefacts are git-ignored): build it once, like
    __graft_entry__.build() does, so the ABI / planner tests (CPU) and every GPU test find it."""
    so = os.path.join(ROOT, "dl-dkd_amd", "dldkd_amd", "libdldkd_hip.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "dl-dkd_amd", "csrc"), "-j4"], check=True)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
