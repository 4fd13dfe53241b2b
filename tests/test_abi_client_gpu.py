"""GPU: the C ABI used from a stand-alone C++ program (tests/c/abi_client.cpp) - no Python, no torch in the
process.  The client packs a small ragged two-branch gallery (streaming packer), scores it and checks the fused
matrix against its own scalar loop."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cxx_client_of_the_c_abi():
    exe = os.path.join(ROOT, "tests", "c", "abi_client")
    if not os.path.exists(exe):          # normally built by __graft_entry__.build()
        subprocess.run(["make", "-C", os.path.join(ROOT, "dl-dkd_amd", "csrc"), "client"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "abi_client ok" in r.stdout
