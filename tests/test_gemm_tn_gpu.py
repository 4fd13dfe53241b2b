"""gemm_bf16_tn.hip (weight gradients as C = A^T B over bf16 rows: LDS-DMA row tiles, transposed LDS reads) and dw_finish_kernel
(split-K reduce + position sums + LayerNorm parameter sums) through the C ABI, against fp64 of the same bf16 operands.
Reference: the backward pass of the towers' Linear / LayerNorm layers (method/model_components.py:277-284, 398-450)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H = 384


def _lib():
    from dldkd_amd import native
    return native, native.lib()


def _dw(rows, blocks, flags=None, dx1=None, n_seq=0, L=0, ln=None):
    """blocks = [(A tensor (rows, lda) bf16 or fp32, acol, B (rows, 384) bf16)] -> dW, dbias[, dpos][, ln_grads]"""
    native, L_ = _lib()
    p = native.ptr
    nb = len(blocks)
    dW = torch.full((nb * H, H), float("nan"), device=DEV)
    dB = torch.zeros(nb * H, device=DEV)
    wsb = L_.dldkd_tower_train_dw_workspace_bytes(nb, rows)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
    hA, hB = (ctypes.c_void_p * nb)(), (ctypes.c_void_p * nb)()
    hl, hc, h16 = (ctypes.c_int * nb)(), (ctypes.c_int * nb)(), (ctypes.c_int * nb)()
    for i, (a, col, b) in enumerate(blocks):
        hA[i], hB[i], hl[i], hc[i], h16[i] = a.data_ptr(), b.data_ptr(), a.shape[1], col, int(a.dtype == torch.bfloat16)
    dpos = torch.zeros(L, H, device=DEV) if dx1 is not None else None
    if ln is not None:
        dz1, xh1, dh2, xh2 = ln
        lnp = torch.zeros(4, H, device=DEV)
        native.check(L_.dldkd_tower_train_dw_ln(hA, hl, hc, h16, hB, nb, rows, p(dW), p(dB), p(ws), wsb, p(flags), p(dx1), p(dpos), n_seq, L * H,
                                                p(dz1), p(xh1), p(dh2), int(dh2.dtype == torch.bfloat16), p(xh2), p(lnp), native.stream()), "dw_ln")
        torch.cuda.synchronize()
        return dW, dB, dpos, lnp
    if dx1 is not None:
        native.check(L_.dldkd_tower_train_dw_pos(hA, hl, hc, h16, hB, nb, rows, p(dW), p(dB), p(ws), wsb, p(flags), p(dx1), p(dpos), n_seq, L * H,
                                                 native.stream()), "dw_pos")
    else:
        native.check(L_.dldkd_tower_train_dw(hA, hl, hc, h16, hB, nb, rows, p(dW), p(dB), p(ws), wsb, p(flags), native.stream()), "dw")
    torch.cuda.synchronize()
    return dW, dB, dpos


def _rel(a, r):
    return ((a.double().cpu() - r).norm() / r.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("rows,use_flags", [(32, False), (96, False), (160, True), (3072, True), (16384, True), (19200, False), (8192 + 32, True)])
def test_grouped_weight_gradients_vs_fp64(rows, use_flags):
    """Five blocks as a video tower issues them: an fp32 A block (cast on the way), a bf16 one, three column ranges of one (rows, 1152)
    bf16 tensor sharing their B; rows of flagged-off 32-row groups hold NaN in EVERY operand and must not be touched; row counts
    off the 64-row tile (a last half tile) and below one tile."""
    g = torch.Generator(device=DEV).manual_seed(rows)
    rnd = lambda *s: torch.randn(*s, generator=g, device=DEV)                        # noqa: E731
    dout, ddo, dqkv = rnd(rows, H), rnd(rows, H).bfloat16(), rnd(rows, 3 * H).bfloat16()
    h2, ctx, h1d = rnd(rows, H).bfloat16(), rnd(rows, H).bfloat16(), rnd(rows, H).bfloat16()
    flags = None
    keep = torch.ones(rows, dtype=torch.bool, device=DEV)
    if use_flags:
        fl = (torch.rand(rows // 32, generator=g, device=DEV) > 0.35)
        fl[0] = True
        keep = fl.repeat_interleave(32)
        flags = fl.to(torch.uint8)
        for t in (dout, ddo, dqkv, h2, ctx, h1d):
            t[~keep] = float("nan")
    blocks = [(dout, 0, h2), (ddo, 0, ctx)] + [(dqkv, c * H, h1d) for c in range(3)]
    dW, dB, _ = _dw(rows, blocks, flags)
    assert torch.isfinite(dW).all() and torch.isfinite(dB).all()
    kd = keep.cpu()
    for i, (a, col, b) in enumerate(blocks):
        a64 = a[:, col:col + H].bfloat16().double().cpu()[kd]                          # the kernel contracts bf16 rows
        ref, refb = a64.t() @ b.double().cpu()[kd], a64.sum(0)
        assert _rel(dW[i * H:(i + 1) * H], ref) <= 2e-6, (i, _rel(dW[i * H:(i + 1) * H], ref))
        eb = _rel(dB[i * H:(i + 1) * H], refb)
        if a.dtype == torch.float32:         # (the register-staged fallback, DLDKD_DW_TN=0, sums the fp32 rows before rounding them)
            eb = min(eb, _rel(dB[i * H:(i + 1) * H], a[:, col:col + H].double().cpu()[kd].sum(0)))
        assert eb <= 2e-6, (i, eb)


@pytest.mark.parametrize("n_seq,L,video", [(128, 128, True), (150, 30, False), (5, 32, True), (3, 20, False)])
def test_finish_launch_position_and_layernorm_sums_vs_fp64(n_seq, L, video):
    """dldkd_tower_train_dw_ln: beside the weight gradients, the position table's gradient (sum of dx1 over the sequences) and both
    LayerNorms' parameter gradients from the rows the backward kernels leave (bf16 dz1 with xh1; dh2 as bf16 under
    an out mapping, else the fp32 rows the loss handed in).  Sequence lengths off the 32-row grid run without flags."""
    rows = n_seq * L
    g = torch.Generator(device=DEV).manual_seed(7 * n_seq + L)
    rnd = lambda *s: torch.randn(*s, generator=g, device=DEV)                        # noqa: E731
    ddo, ctx = rnd(rows, H).bfloat16(), rnd(rows, H).bfloat16()
    dx1 = rnd(n_seq, L * H)
    dz1, dh2 = rnd(rows, H).bfloat16(), (rnd(rows, H).bfloat16() if video else rnd(rows, H))
    xh2 = rnd(rows, H).bfloat16()
    xh1 = (rnd(rows, H).bfloat16().view(torch.int16) | torch.randint(0, 2, (rows, H), generator=g, device=DEV).to(torch.int16)).view(torch.bfloat16)   # (every bit a value bit: the ReLU mask left bit 0 in round 6)
    flags, keep = None, torch.ones(rows, dtype=torch.bool, device=DEV)
    if rows % 32 == 0 and L % 32 == 0:
        fl = torch.rand(rows // 32, generator=g, device=DEV) > 0.3
        fl[0] = True
        flags, keep = fl.to(torch.uint8), fl.repeat_interleave(32)
        for t in (ddo, ctx, dz1, dh2, xh1, xh2):
            t[~keep] = float("nan")
    dW, dB, dpos, lnp = _dw(rows, [(ddo, 0, ctx)], flags, dx1=dx1, n_seq=n_seq, L=L, ln=(dz1, xh1, dh2, xh2))
    kd = keep.cpu()
    assert _rel(dW, ddo.double().cpu()[kd].t() @ ctx.double().cpu()[kd]) <= 2e-6
    assert _rel(dpos, dx1.double().cpu().view(n_seq, L, H).sum(0)) <= 2e-6
    x1 = xh1.double().cpu()[kd]
    ref = [(dh2.double().cpu()[kd] * xh2.double().cpu()[kd]).sum(0), dh2.double().cpu()[kd].sum(0),
           (dz1.double().cpu()[kd] * x1).sum(0), dz1.double().cpu()[kd].sum(0)]
    assert torch.isfinite(lnp).all()
    for i in range(4):
        assert _rel(lnp[i], ref[i]) <= 5e-6, (i, _rel(lnp[i], ref[i]))


def test_register_staged_fallback_for_shapes_off_the_grid():
    """Row counts that are not a multiple of 32 (no flag groups) go through the register-staged kernel: same results."""
    rows = 1000
    g = torch.Generator(device=DEV).manual_seed(3)
    a, b = torch.randn(rows, H, generator=g, device=DEV).bfloat16(), torch.randn(rows, H, generator=g, device=DEV).bfloat16()
    dW, dB, _ = _dw(rows, [(a, 0, b)])
    assert _rel(dW, a.double().cpu().t() @ b.double().cpu()) <= 2e-6 and _rel(dB, a.double().cpu().sum(0)) <= 2e-6


@pytest.mark.parametrize("M,K,p_drop", [(4096, 3072, 0.2), (1056, 768, 0.0), (640, 1024, 0.1)])
def test_input_projection_backward_with_and_without_the_bf16_copy_of_dy(M, K, p_drop):
    """dldkd_inproj_bwd_bf16's dy_bf16 argument (the rows dldkd_tower_train_b1 leaves beside dy0): the same planes as when the
    library casts dy itself - bit for bit for dW, the LayerNorm sums and the bias up to their fp32 atomics - and dW against fp64."""
    native, L_ = _lib()
    p = native.ptr
    N = H
    g = torch.Generator(device=DEV).manual_seed(M + K)
    rnd = lambda *s: torch.randn(*s, generator=g, device=DEV)                        # noqa: E731
    x = rnd(M, K)
    gamma, beta, W = 1 + 0.1 * rnd(K), 0.1 * rnd(K), 0.03 * rnd(N, K)
    z = torch.empty(M, K, dtype=torch.bfloat16, device=DEV)
    stats = torch.empty(2, M, device=DEV)
    native.check(L_.dldkd_layernorm_dropout_bf16(p(x), p(gamma), p(beta), p(z), None, p(stats), M, K, 1e-5, p_drop, 11, 0, None, None, None,
                                                 native.stream()), "ln")
    dy = rnd(M, N)
    dy16 = dy.bfloat16()
    nbytes = L_.dldkd_inproj_bwd_workspace_bytes(N, K, M)
    res = []
    for given in (None, dy16):
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        dW, db, dgb = torch.empty(N, K, device=DEV), torch.zeros(N, device=DEV), torch.zeros(2, K, device=DEV)
        native.check(L_.dldkd_inproj_bwd_bf16(p(dy), p(z), p(W), p(gamma), p(beta), 1.0 / (1.0 - p_drop), p(x), None, p_drop, 11, 0, None,
                                              p(stats[0]), p(stats[1]), p(dW), p(db), p(dgb[0]), p(dgb[1]), M, N, K, p(ws), nbytes, None,
                                              p(given), native.stream()), "inproj_bwd")
        torch.cuda.synchronize()
        res.append((dW, db, dgb))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.allclose(res[0][2], res[1][2], rtol=1e-5, atol=1e-6 * float(res[0][2].abs().max()))
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-5, atol=1e-6 * float(res[0][1].abs().max()))
    assert _rel(res[0][0], dy16.double().cpu().t() @ z.double().cpu()) <= 2e-6
    assert _rel(res[0][1], dy16.double().cpu().sum(0)) <= 2e-6
