"""CPU: the C-ABI library loads and exports exactly the symbols include/dldkd_hip.h declares
(no compute calls - there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "dldkd_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dldkd_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    from dldkd_amd import native
    if not os.path.exists(native.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = native.lib()
    names = _declared()
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dldkd_hip.h but not exported"
        assert n in native.SIGNATURES, f"{n} has no ctypes signature in native.py"
    assert sorted(native.SIGNATURES) == names
    assert lib.dldkd_abi_version() == native.ABI_VERSION


def test_size_helpers_and_argument_checks():
    from dldkd_amd import native
    lib = native.lib()
    assert lib.dldkd_packed_queries_bytes(1) == 32 * 384 * 2
    assert lib.dldkd_packed_queries_bytes(33) == 64 * 384 * 2
    assert lib.dldkd_packed_gallery_bytes(5, 9) == 5 * 32 * 384 * 2
    assert lib.dldkd_packed_gallery_bytes(3, 128) == 3 * 128 * 384 * 2
    assert lib.dldkd_simpool_eval_workspace_bytes(33, 7, 2) == 2 * 7 * 64 * 4
    # argument validation happens before any HIP call, so it is testable without a GPU
    assert lib.dldkd_pack_gallery_bf16(None, None, 4, 129, 1, None, None, None) == -1
    assert b"L must be" in lib.dldkd_last_error()
    assert lib.dldkd_simpool_eval_bf16(None, None, None, None, 5, 5, 16, 3, None, None) == -1


def test_no_cpu_fallback():
    """The product path refuses CPU tensors instead of silently computing somewhere else."""
    import torch
    from dldkd_amd import native, scoring
    with pytest.raises(native.NativeError):
        scoring.pack_queries([torch.zeros(4, 384)])


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "dl-dkd_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "dldkd_oracle" not in src and "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S), f


def test_header_is_plain_c():
    """The boundary is a C ABI: the header must compile as C99 (no C++-only constructs outside extern "C" guards)."""
    import subprocess
    hdr = os.path.join(ROOT, "include", "dldkd_hip.h")
    r = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
