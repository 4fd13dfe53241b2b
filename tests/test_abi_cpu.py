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
    assert lib.dldkd_simpool_eval_bf16(None, None, None, None, 5, 5, 16, 3, 0, None, None, None) == -1


def test_query_split_plan_host():
    """dldkd_simpool_eval_plan is pure host code: the round-count model behind the scorer's [range][branch][group] grid."""
    from dldkd_amd import scoring
    assert scoring.plan_query_split(10895, 21793, 2) == (1, 10912)              # C2 on one GPU: 42.6 rounds, no split
    n, per = scoring.plan_query_split(17505, 615, 2)                            # C4, one of 8 shards: 308 workgroups
    rounds = -(-308 * n // 256)
    assert n >= 3 and per % 32 == 0 and 308 * n / (256 * rounds) >= 0.8
    assert (n - 1) * per < 17505 <= n * per
    n8, per8 = scoring.plan_query_split(17505, 615, 2, min_split=8)
    assert n8 >= 8 and (n8 - 1) * per8 < 17505 <= n8 * per8
    assert scoring.plan_query_split(5, 3, 1, min_split=4) == (1, 32)            # one tile cannot be split
    for nq, nv in ((1, 1), (33, 7), (4096, 64), (17505, 4917), (10895, 2725)):
        for nb in (1, 2):
            n, per = scoring.plan_query_split(nq, nv, nb)
            assert n >= 1 and per % 32 == 0 and (n - 1) * per < max(nq, 1) <= n * per


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


def test_library_only_enqueues():
    """include/dldkd_hip.h promises: no allocation, no free, no host synchronisation, no blocking copy anywhere in the
    library (a process-global split-K buffer used to hipMalloc / hipDeviceSynchronize / hipFree inside a
    GEMM call).  Checked on the sources the .so is built from."""
    csrc = os.path.join(ROOT, "dl-dkd_amd", "csrc")
    banned = re.compile(r"\b(hipMalloc\w*|hipFree\w*|hipHostMalloc|hipDeviceSynchronize|hipStreamSynchronize|hipEventSynchronize|"
                        r"hipMemcpy|hipMemcpyDtoH|hipMemcpyHtoD|hipMemset|hipExtMallocWithFlags)\s*\(")
    seen = 0
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".hpp", ".h", ".cpp")):
            src = open(os.path.join(csrc, f)).read()
            src = re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", src, flags=re.S))
            m = banned.search(src)
            assert m is None, f"{f}: {m.group(0)} in the enqueue-only library"
            seen += 1
    assert seen >= 10


def test_gemm_workspace_bytes_host():
    """Pure host code: forward layouts never split K; the weight-gradient shapes of the C3 / C5 steps do."""
    from dldkd_amd import native
    lib = native.lib()
    for prec in (0, 1, 2):
        assert lib.dldkd_gemm_workspace_bytes(prec, 16384, 384, 3072, 0, 0) == 0          # Linear forward
        assert lib.dldkd_gemm_workspace_bytes(prec, 0, 384, 3072, 1, 1) == 0
        n = lib.dldkd_gemm_workspace_bytes(prec, 384, 3072, 16384, 1, 1)                   # dW of the input projection
        assert n >= 2 * 384 * 3072 * 4 and n % (384 * 3072 * 4) == 0
        assert lib.dldkd_gemm_workspace_bytes(prec, 16384, 3072, 384, 0, 1) == 0          # dX: large output grid, short K
    assert lib.dldkd_gemm_workspace_bytes(7, 384, 3072, 16384, 1, 1) == 0
