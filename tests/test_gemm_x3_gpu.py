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


@pytest.mark.parametrize("M,N,K", [(1, 1, 4), (130, 70, 20), (257, 384, 3072), (1024, 1152, 384), (64, 384, 770)])
def test_x2_forward_is_two_plane_grade(M, N, K):
    """The two-plane form (ops "fp32x2", the forward GEMMs of the "mixed" training precision): ~2^-16 per product, i.e. between the
    three-plane kernel (2e-6 at these shapes) and bf16 (4e-3); same bias / ReLU / row-flag semantics as the three-plane entry."""
    from dldkd_amd import ops
    old = ops.precision_mode()
    ops.set_gemm_precision("fp32x2")
    try:
        g = torch.Generator().manual_seed(M + N + K)
        x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05, torch.randn(N, generator=g)
        ref = x.double() @ w.double().t() + b.double()
        y = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), relu=True)
        e2 = _rel(y, torch.relu(ref))
        assert e2 < 4e-5, e2
        if M % 128 == 0:                                       # row flags: groups flagged 0 are not multiplied (their rows: act(bias))
            flags = torch.ones(M // 32, dtype=torch.uint8, device=DEV)
            flags[1::4] = 0
            yf = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), relu=True, row_flags=flags)
            keep = flags.bool().repeat_interleave(32).cpu()
            assert torch.equal(yf.cpu()[keep], y.cpu()[keep])
            assert torch.allclose(yf.cpu()[~keep], torch.relu(b).expand(int((~keep).sum()), N))
        ops.set_gemm_precision("fp32x3")
        e3 = _rel(ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), relu=True), torch.relu(ref))
        assert e3 <= e2 or e2 < 1e-6
    finally:
        ops.set_gemm_precision(old)


@pytest.mark.parametrize("M,N,K", [(2048, 384, 3072), (1024, 1152, 384), (300, 384, 768), (128, 70, 64)])
def test_two_plane_gemm_over_plane_segments_matches_the_in_kernel_split(M, N, K):
    """dldkd_gemm_bf16_nt16_planes (the LDS-DMA bf16 kernel over three K-long segments of pre-split operands: functional.split2_jobs /
    gemm_planes) computes the two-plane product of dldkd_gemm_f32x2: same error class against fp64 (4e-5), row flags honoured, and
    dldkd_split2_bf16_jobs writes h = bf16(x), m = bf16(x - h) (x = h + m to 2^-16) and gathers fp32 vectors."""
    from dldkd_amd import functional as F_, ops
    old = ops.precision_mode()
    ops.set_gemm_precision("fp32x2")
    try:
        g = torch.Generator().manual_seed(M + N + K)
        x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05, torch.randn(N, generator=g)
        xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
        b3 = [torch.randn(4, generator=g).to(DEV) for _ in range(3)]
        xp, wp, bcat = F_.split2_jobs([(xd, "split"), (wd, "split"), (tuple(b3), "copy")], xd.device)
        assert xp.shape == (2, M, K) and wp.shape == (2, N, K) and torch.equal(bcat, torch.cat(b3))
        assert torch.equal(xp[0], xd.bfloat16()) and torch.equal(xp[1], (xd - xd.bfloat16().float()).bfloat16())
        assert float((xp[0].float() + xp[1].float() - xd).abs().max()) <= 2.0 ** -15 * float(xd.abs().max())
        ref = torch.relu(x.double() @ w.double().t() + b.double())
        y = F_.gemm_planes(xp, wp, bd, M, N, K, relu=True)
        assert _rel(y, ref) < 4e-5
        y2 = ops.linear(xd, wd, bd, relu=True)                      # the in-kernel-split two-plane kernel
        assert _rel(y, y2.double().cpu()) < 2e-6                    # same products, another summation order
        if M % 128 == 0:
            flags = torch.ones(M // 32, dtype=torch.uint8, device=DEV)
            flags[2::4] = 0
            yf = F_.gemm_planes(xp, wp, bd, M, N, K, relu=True, row_flags=flags)
            keep = flags.bool().repeat_interleave(32).cpu()
            assert torch.equal(yf.cpu()[keep], y.cpu()[keep])
    finally:
        ops.set_gemm_precision(old)
