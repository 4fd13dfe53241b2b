"""GPU: the C ABI's "enqueue only" contract (include/dldkd_hip.h, Conventions): a split-K weight-gradient GEMM
(dW = dY^T X, method/train.py:141-151's backward) is captured into a hipGraph and replayed - legal only because the library
never allocates, frees or synchronises (the workspace is the caller's) - and a call without a workspace is still correct."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("prec", ["fp32", "fp32_exact", "bf16"])
def test_splitk_dw_gemm_captures_into_a_hip_graph(prec):
    from dldkd_amd import native, ops
    old = ops.gemm_precision()
    ops.set_gemm_precision(prec)
    try:
        L = native.lib()
        M, N, K = 384, 3072, 16384                   # dW of the visual input projection at C3: [384 x 3072] over 16,384 rows
        assert L.dldkd_gemm_workspace_bytes(ops._PREC_ID[prec], M, N, K, 1, 1) >= 2 * M * N * 4      # this shape splits
        g = torch.Generator(device=DEV).manual_seed(3)
        dy = torch.randn(K, M, generator=g, device=DEV)
        x = torch.randn(K, N, generator=g, device=DEV)
        eager = ops.gemm(dy, x, True, True, M, N, K)
        ref = dy.double().t() @ x.double()
        tol = 2e-2 if prec == "bf16" else 2e-5          # fp32 accumulation over K = 16384 terms (tests/test_gemm_x3_gpu.py scales the same way)
        assert ((eager.double() - ref).abs().max() / ref.abs().max()).item() < tol
        # no workspace: legal, unsplit, same numbers up to summation order
        c = torch.empty(M, N, device=DEV)
        fn = ops._gemm_fn(L)
        native.check(fn(native.ptr(dy), native.ptr(x), None, native.ptr(c), M, N, K, M, N, N, 1, 1, 0, None, 0, native.stream()), "gemm")
        assert ((c.double() - ref).abs().max() / ref.abs().max()).item() < tol
        # capture + replay on fresh inputs written into the captured buffers
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            ops.gemm(dy, x, True, True, M, N, K)       # warm-up on the capture stream
            s.synchronize()
            with torch.cuda.graph(graph, stream=s):
                out = ops.gemm(dy, x, True, True, M, N, K)
        torch.cuda.current_stream().wait_stream(s)
        dy.copy_(torch.randn(K, M, generator=g, device=DEV))
        x.copy_(torch.randn(K, N, generator=g, device=DEV))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ops.gemm(dy, x, True, True, M, N, K))      # split-K has a fixed reduce order: bitwise equal
    finally:
        ops.set_gemm_precision(old)


def test_two_streams_do_not_share_scratch():
    """Round 1 kept ONE process-global split-K buffer: two streams running weight-gradient GEMMs clobbered each other's
    partial planes.  Scratch now comes from the caller per call."""
    from dldkd_amd import ops
    M, N, K = 384, 384, 32768
    g = torch.Generator(device=DEV).manual_seed(5)
    a = [torch.randn(K, M, generator=g, device=DEV) for _ in range(2)]
    b = [torch.randn(K, N, generator=g, device=DEV) for _ in range(2)]
    ref = [ops.gemm(a[i], b[i], True, True, M, N, K) for i in range(2)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for _ in range(5):
        outs = [None, None]
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                for _ in range(4):
                    outs[i] = ops.gemm(a[i], b[i], True, True, M, N, K)
        torch.cuda.synchronize()
        for i in range(2):
            assert torch.equal(outs[i], ref[i])
