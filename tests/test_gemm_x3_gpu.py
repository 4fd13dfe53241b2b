"""GPU: the fp32-grade three-plane bf16 GEMM (gemm_f32x3.hip) against fp64, at the SAME tolerances as the true
fp32-input-MFMA kernel (tests/test_encoder_gpu.py): it has to be a drop-in for the parity path."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture
def x3():
    from dldkd_amd import ops
    old = ops.gemm_precision()
    ops.set_gemm_precision("fp32x3")
    yield
    ops.set_gemm_precision(old)


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("M,N,K", [(1, 1, 4), (130, 70, 20), (257, 384, 3072), (1000, 1152, 384), (64, 384, 770)])
def test_x3_linear_forward(x3, M, N, K):
    from dldkd_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05, torch.randn(N, generator=g)
    y = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), relu=True)
    assert _rel(y, torch.relu(x.double() @ w.double().t() + b.double())) < 2e-6
    y2 = ops.linear(x.to(DEV), w.to(DEV))
    assert _rel(y2, x.double() @ w.double().t()) < 2e-6


@pytest.mark.parametrize("M,N,K", [(96, 50, 36), (300, 384, 384), (640, 384, 16384), (129, 3072, 385)])
def test_x3_backward_layouts(x3, M, N, K):
    from dldkd_amd import ops
    g = torch.Generator().manual_seed(7)
    dy, w, x = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g), torch.randn(M, N, generator=g)
    dx = ops.gemm(dy.to(DEV), w.to(DEV), False, True, M, N, K)
    assert _rel(dx, dy.double() @ w.double()) < 2e-6 * max(1.0, (K / 512) ** 0.5)
    Kc = 128 if K > 128 else K
    dyc = dy[:, :Kc].contiguous()
    dw = ops.gemm(dyc.to(DEV), x.to(DEV), True, True, Kc, N, M)
    assert _rel(dw, dyc.double().t() @ x.double()) < 2e-6


def test_x3_error_is_fp32_grade_not_bf16_grade(x3):
    """Elementwise: against fp64 the three-plane product must be as good as a true fp32 GEMM (and ~1000x better than bf16)."""
    from dldkd_amd import ops
    g = torch.Generator().manual_seed(1)
    x, w = torch.randn(512, 768, generator=g), torch.randn(384, 768, generator=g)
    ref = x.double() @ w.double().t()
    y3 = ops.linear(x.to(DEV), w.to(DEV)).cpu().double()
    ops.set_gemm_precision("fp32_exact")
    y1 = ops.linear(x.to(DEV), w.to(DEV)).cpu().double()
    ops.set_gemm_precision("bf16")
    yb = ops.linear(x.to(DEV), w.to(DEV)).cpu().double()
    e3, e1, eb = (y3 - ref).abs().max().item(), (y1 - ref).abs().max().item(), (yb - ref).abs().max().item()
    assert e3 < 4 * e1 + 1e-6 and e3 < eb / 200, (e3, e1, eb)
