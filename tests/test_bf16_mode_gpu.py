"""Throughput mode of the training step: the GEMMs on bf16 MFMA (fp32 accumulate), everything else as in fp32.

The bf16 GEMM is checked exactly (against fp64 products of the bf16-rounded operands: only accumulation order
differs) in every operand layout the training step uses; the whole forward+backward is then checked against the
reference's golden losses/gradients (G4) at bf16-grade tolerance.  fp32 remains the parity mode (test_train_gpu)."""
import numpy as np
import pytest
import torch

import synth
from test_encoder_gpu import _model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture
def bf16_mode():
    from dldkd_amd import ops
    ops.set_gemm_precision("bf16")
    yield
    ops.set_gemm_precision("fp32")


def _r(t):
    return t.bfloat16().double()


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("M,N,K", [(1, 1, 4), (130, 70, 20), (257, 384, 3072), (1000, 1152, 384), (64, 384, 770)])
def test_gemm_bf16_linear(bf16_mode, M, N, K):
    from dldkd_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05, torch.randn(N, generator=g)
    y = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), relu=True)
    assert _rel(y, torch.relu(_r(x) @ _r(w).t() + b.double())) < 3e-6
    assert _rel(y, torch.relu(x.double() @ w.double().t() + b.double())) < 2e-2


@pytest.mark.parametrize("M,N,K", [(96, 50, 36), (300, 384, 384), (640, 384, 16384), (129, 3072, 385)])
def test_gemm_bf16_backward_layouts(bf16_mode, M, N, K):
    from dldkd_amd import ops
    g = torch.Generator().manual_seed(7)
    dy, w, x = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g), torch.randn(M, N, generator=g)
    dx = ops.gemm(dy.to(DEV), w.to(DEV), False, True, M, N, K)
    assert _rel(dx, _r(dy) @ _r(w)) < 3e-6 * max(1.0, (K / 512) ** 0.5)
    Kc = 128 if K > 128 else K
    dyc = dy[:, :Kc].contiguous()
    dw = ops.gemm(dyc.to(DEV), x.to(DEV), True, True, Kc, N, M)
    assert _rel(dw, _r(dyc).t() @ _r(x)) < 3e-6


def test_forward_backward_bf16_vs_golden_g4(bf16_mode, golden_dir):
    g = np.load(f"{golden_dir}/g4_forward.npz")
    tag = "soft_rand"
    m = _model(3072, 768, synth.make_params(41, 3072, 768))
    m.label_style = "soft"
    m.set_hard_negative(False, 20)
    m.weight = 0.95 ** 2
    batch = synth.make_train_batch(1, nv=64, caps=1, L=16, dv=3072, dq=768)
    batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    torch.manual_seed(4242)
    loss, d = m(batch)
    for k in ("inher_trip", "inher_nce", "explore_trip", "explore_nce", "kl_intra"):
        ref = float(g[f"{tag}_{k}"])
        assert abs(float(d[k]) - ref) <= 2e-2 * max(1.0, abs(ref)), (k, float(d[k]), ref)
    ref_loss = float(np.asarray(g[f"{tag}_loss"]).reshape(-1)[0])
    assert abs(float(loss) - ref_loss) <= 2e-2 * max(1.0, abs(ref_loss))
    m.zero_grad()
    loss.backward()
    # direction and size of the whole gradient: cosine over the sampled entries, norms per tensor
    got, ref, bad = [], [], []
    nmax = max(float(g[f"{tag}_grad/{n}/norm"]) for n, _ in m.named_parameters())
    for n, prm in m.named_parameters():
        gr = prm.grad.detach().reshape(-1).cpu()
        idx = np.unique(np.linspace(0, gr.numel() - 1, min(48, gr.numel())).astype(np.int64))
        got.append(gr[idx].double().numpy())
        ref.append(g[f"{tag}_grad/{n}/sample"].astype(np.float64))
        rn = float(g[f"{tag}_grad/{n}/norm"])
        if abs(float(gr.double().norm()) - rn) > 0.1 * max(rn, 1e-3 * nmax):
            bad.append((n, float(gr.double().norm()), rn))
    got, ref = np.concatenate(got), np.concatenate(ref)
    cos = float(got @ ref / (np.linalg.norm(got) * np.linalg.norm(ref)))
    assert cos > 0.97, cos          # measured 0.987: bf16 operand rounding through 8 stacked GEMMs + hinge/ReLU boundaries
    assert not bad, bad


def _h(t):
    """round to the eval-path towers' operand format (h16 = IEEE fp16, csrc/common.hpp)"""
    return t.float().half().double()


@pytest.mark.parametrize("M,K,n_lin,relu", [(1, 384, 1, False), (130, 384, 3, False), (25600, 384, 3, False), (777, 384, 2, True),
                                            (1000, 768, 1, True), (129, 32, 3, False)])
def test_linear_rows_full_row_kernel(bf16_mode, M, K, n_lin, relu):
    """Full-row kernel of the inference chain (weights in MFMA fragment order, fp16 operands like every eval-path tower kernel)
    against fp64 products of the fp16-rounded operands."""
    from dldkd_amd import ops
    torch.manual_seed(M + K + n_lin)
    lins = [torch.nn.Linear(K, 384).to(DEV) for _ in range(n_lin)]
    x = torch.randn(M, K, device=DEV)
    pk = ops.PackedLinear(lins)
    with torch.no_grad():
        y = ops.linear_rows(x, pk, relu=relu)
        ref = torch.cat([_h(x.cpu()) @ _h(l.weight.detach().cpu()).t() + l.bias.detach().cpu().double() for l in lins], 1)
        if relu:
            ref = torch.relu(ref)
        assert y.shape == (M, 384 * n_lin)
        assert _rel(y, ref) < 3e-6
        # parameter update -> repack
        lins[0].weight.mul_(2.0)
        y2 = ops.linear_rows(x, pk, relu=relu)
        ref2 = _h(x.cpu()) @ _h(lins[0].weight.detach().cpu()).t() + lins[0].bias.detach().cpu().double()
        assert _rel(y2[:, :384], torch.relu(ref2) if relu else ref2) < 3e-6


def test_towers_throughput_mode_close_to_parity_mode(bf16_mode):
    """encode_context / encode_query with K4 + full-row bf16 linears vs the fp32 towers: bf16-grade agreement."""
    from dldkd_amd import ops
    m = _model(3072, 768, synth.make_params(41, 3072, 768))
    b = synth.make_train_batch(1, nv=24, caps=2, L=40, dv=3072, dq=768)
    v, vm, t, tm = (b[k].to(DEV) for k in ("student_videos", "student_videos_mask", "student_text", "student_text_mask"))
    with torch.no_grad():
        ops.set_gemm_precision("fp32")
        m.fast_input_proj = False
        ref = list(m.encode_context(v, vm)) + list(m.encode_query(t, tm))
        ops.set_gemm_precision("bf16")
        m.fast_input_proj = True
        got = list(m.encode_context(v, vm)) + list(m.encode_query(t, tm))
    for a, r in zip(got, ref):
        assert a.shape == r.shape
        assert _rel(a, r) < 3e-2


@pytest.mark.parametrize("N,L", [(3, 128), (2, 1), (5, 33), (4, 100), (7, 30)])
def test_attention_bf16_vs_reference_math(bf16_mode, N, L):
    """bf16-MFMA fused attention against fp64 softmax(QK^T/sqrt(96) + (1-mask)*-1e4) V on the bf16-rounded q|k|v
    (the probabilities are additionally rounded to bf16 inside the kernel: 2^-9 relative)."""
    from dldkd_amd import ops
    g = torch.Generator().manual_seed(N * 131 + L)
    qkv = torch.randn(N, L, 1152, generator=g)
    lens = torch.randint(1, L + 1, (N,), generator=g)
    lens[0] = L
    mask = (torch.arange(L)[None] < lens[:, None]).float()
    with torch.no_grad():
        out = ops.attention(qkv.to(DEV), mask.to(DEV)).cpu().double()
        assert ops.gemm_precision() == "bf16"
    x = _r(qkv).view(N, L, 3, 4, 96)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)     # (N, 4, L, 96)
    s = q @ k.transpose(-1, -2) / 96 ** 0.5 + (1.0 - mask.double())[:, None, None, :] * -10000.0
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(N, L, 384)
    assert (out - ref).abs().max().item() < 6e-3 * ref.abs().max().item() + 1e-6
    none = ops.attention(qkv.to(DEV), None).cpu().double()
    s2 = q @ k.transpose(-1, -2) / 96 ** 0.5
    ref2 = (torch.softmax(s2, -1) @ v).transpose(1, 2).reshape(N, L, 384)
    assert (none - ref2).abs().max().item() < 6e-3 * ref2.abs().max().item() + 1e-6


def test_eval_epoch_throughput_mode_vs_parity_mode(golden_dir):
    """The whole eval path with K4 + full-row bf16 linears + bf16 attention against the fp32-tower path on the G5
    inputs: score matrices agree to bf16 grade, R@K moves by at most two of the 192 queries per cut (random-init
    weights give near-ties), and both stay within that distance of the REFERENCE's own R@K (golden G5)."""
    import types
    from dldkd_amd import eval as ev, ops
    g = np.load(f"{golden_dir}/g5_eval_epoch.npz")
    m = _model(3072, 768, synth.make_params(51, 3072, 768))
    vids, txts = synth.make_eval_sets(5, nv=64, caps=3, dv=3072, dq=768)
    opt = types.SimpleNamespace(eval_context_bsz=25, eval_query_bsz=50, num_workers=0, pin_memory=False,
                                device=torch.device(DEV), double_branch=True)
    res = {}
    try:
        for mode in ("fp32", "bf16"):
            ops.set_gemm_precision(mode)
            m.fast_input_proj = mode == "bf16"
            with torch.no_grad():
                ctx = ev.compute_context_info(m, synth.ListDataset(list(vids)), opt, keep_frame_feats=False)
                fused, s0, s1, qmetas = ev.score_queries(m, synth.ListDataset(list(txts)), opt, ctx)
            _, t2v = ev.get_gt(ctx["video_metas"], qmetas)
            res[mode] = (fused.cpu(), ev.eval_q2m(-fused, t2v))
    finally:
        ops.set_gemm_precision("fp32")
        m.fast_input_proj = False
    assert (res["bf16"][0] - res["fp32"][0]).abs().max().item() < 2e-2
    step = 100.0 / 192
    for a, b, r in zip(res["bf16"][1][:4], res["fp32"][1][:4], g["perf_fused"][:4]):
        assert abs(a - b) <= 2 * step + 1e-9 and abs(a - r) <= 2 * step + 1e-9, (res["bf16"][1], res["fp32"][1], g["perf_fused"])


def test_bf16_qkv_handoff_equals_fp32_handoff(bf16_mode):
    """linear_rows(out_bf16) -> attention(bf16 qkv) is the same computation as fp32 qkv -> attention (which rounds
    q|k|v to bf16 itself): bit-identical outputs."""
    from dldkd_amd import ops
    torch.manual_seed(5)
    lins = [torch.nn.Linear(384, 384).to(DEV) for _ in range(3)]
    pk = ops.PackedLinear(lins)
    for shape in ((7, 50, 384), (3, 128, 384), (1, 1, 384)):
        x = torch.randn(*shape, device=DEV)
        mask = torch.ones(shape[0], shape[1], device=DEV)
        if shape[1] > 20:
            mask[0, 20:] = 0
        with torch.no_grad():
            q32 = ops.linear_rows(x, pk)
            q16 = ops.linear_rows(x, pk, out_bf16=True)
            assert q16.dtype == torch.bfloat16 and torch.equal(q16.float(), q32.bfloat16().float())
            assert torch.equal(ops.attention(q32, mask), ops.attention(q16, mask))


def test_packed_weight_caches_follow_the_fused_optimizer(golden_dir):
    """the throughput-mode towers run on packed bf16 copies of the weights (FoldedInProj, PackedLinear)
    keyed on (data_ptr, _version); the fused BertAdam step writes parameters through raw pointers and changes neither.
    eval -> native optimizer steps -> eval must equal a FRESH model built from the same state_dict."""
    import types
    from dldkd_amd import eval as ev, ops
    from dldkd_amd.optimization import BertAdam
    m = _model(3072, 768, synth.make_params(51, 3072, 768))
    vids, txts = synth.make_eval_sets(5, nv=24, caps=2, dv=3072, dq=768)
    opt = types.SimpleNamespace(eval_context_bsz=25, eval_query_bsz=50, num_workers=0, pin_memory=False,
                                device=torch.device(DEV), double_branch=True)

    def scores(model, fast=True):
        model.eval()
        model.fast_input_proj = fast
        with torch.no_grad():
            ctx = ev.compute_context_info(model, synth.ListDataset(list(vids)), opt, keep_frame_feats=False)
            return ev.score_queries(model, synth.ListDataset(list(txts)), opt, ctx)[0].clone()
    before_parity = scores(m, fast=False)                          # parity inference: FoldedInProjX3 / PackedLinearX3 planes
    ops.set_gemm_precision("bf16")
    try:
        before = scores(m)                                         # builds every packed cache
        optim = BertAdam([{"params": list(m.parameters()), "weight_decay": 0.01}], lr=5e-3, warmup=-1, t_total=-1, schedule="none")
        m.train()
        batch = synth.make_train_batch(3, nv=16, caps=2, L=12, dv=3072, dq=768)
        batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
        for _ in range(3):
            optim.zero_grad()
            loss, _ = m(batch)
            loss.backward()
            optim.step()
        after = scores(m)
        fresh = _model(3072, 768, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
        want = scores(fresh)
    finally:
        ops.set_gemm_precision("fp32")
    assert (after - before).abs().max().item() > 1e-3              # the steps changed the scores at all
    assert torch.equal(after, want)                                # and eval sees the CURRENT weights, exactly
    after_parity, want_parity = scores(m, fast=False), scores(fresh, fast=False)   # same for the three-plane caches of parity mode
    assert (after_parity - before_parity).abs().max().item() > 1e-3
    assert torch.equal(after_parity, want_parity)


@pytest.mark.parametrize("M,N,K,p_drop", [(300, 384, 3072, 0.2), (129, 384, 776, 0.0), (1000, 130, 100, 0.5), (64, 2, 36, 0.1)])
def test_bf16_rows_of_the_training_input_projection(bf16_mode, M, N, K, p_drop):
    """dldkd_layernorm_dropout_bf16 + dldkd_gemm_bf16_mixed (functional._InProjTrain in throughput mode: the LayerNorm-dropout
    rows are stored as bf16, the forward GEMM and dW read them as they are) against the fp32-row kernels they replace: the rows
    are the bf16 rounding of dldkd_layernorm_dropout_f32's (same masks), the statistics those of dldkd_row_meanrstd_f32, and
    both GEMMs give the numbers dldkd_gemm_bf16 gives on the fp32 rows (which it rounds to bf16 itself) - ragged edge tiles,
    K not a multiple of the k-tile, split-K dW included."""
    from dldkd_amd import native, ops
    L = native.lib()
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * (1 + torch.rand(M, 1, generator=g)) + 0.2).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(K, generator=g)).to(DEV), (0.1 * torch.randn(K, generator=g)).to(DEV)
    W, b = (torch.randn(N, K, generator=g) * 0.05).to(DEV), (0.1 * torch.randn(N, generator=g)).to(DEV)
    dy = torch.randn(M, N, generator=g).to(DEV)
    z32, k32 = torch.empty(M, K, device=DEV), torch.empty(M, K, dtype=torch.uint8, device=DEV)
    z16, k16 = torch.empty(M, K, dtype=torch.bfloat16, device=DEV), torch.empty(M, K, dtype=torch.uint8, device=DEV)
    stats, ref_stats = torch.empty(2, M, device=DEV), torch.empty(2, M, device=DEV)
    if p_drop > 0:
        native.check(L.dldkd_layernorm_dropout_f32(native.ptr(x), None, 0, native.ptr(gamma), native.ptr(beta), native.ptr(z32), native.ptr(k32),
                                                   M, K, 1e-5, p_drop, 1234, 40, None, native.stream()), "ln")
    else:
        z32 = ops.layernorm(x, gamma, beta)
    native.check(L.dldkd_layernorm_dropout_bf16(native.ptr(x), native.ptr(gamma), native.ptr(beta), native.ptr(z16),
                                                native.ptr(k16) if p_drop > 0 else None, native.ptr(stats), M, K, 1e-5, p_drop, 1234, 40, None,
                                                None, None, native.stream()), "ln16")
    native.check(L.dldkd_row_meanrstd_f32(native.ptr(x), native.ptr(ref_stats[0]), native.ptr(ref_stats[1]), M, K, 1e-5, native.stream()), "stats")
    assert torch.equal(z16, z32.to(torch.bfloat16)) and torch.equal(stats, ref_stats)
    if p_drop > 0:
        assert torch.equal(k16, k32) and 0.5 * (1 - p_drop) < k16.float().mean().item() < 1 - 0.5 * p_drop
    y = torch.empty(M, N, device=DEV)
    native.check(L.dldkd_gemm_bf16_mixed(0, native.ptr(z16), native.ptr(W), native.ptr(b), native.ptr(y), M, N, K, K, K, N, 1, None, 0, None,
                                         native.stream()), "fwd")
    assert torch.equal(y, ops.linear(z32, W, b, relu=True))
    dw = torch.empty(N, K, device=DEV)
    ws, nb = ops._gemm_workspace(L, N, K, M, True, True, x.device)
    native.check(L.dldkd_gemm_bf16_mixed(1, native.ptr(dy), native.ptr(z16), None, native.ptr(dw), N, K, M, N, K, K, 0, native.ptr(ws), nb, None,
                                         native.stream()), "dw")
    ref = ops.gemm(dy, z32, True, True, N, K, M)
    assert torch.allclose(dw, ref, rtol=1e-5, atol=1e-6 * ref.abs().max().item())        # (split-K planes: fp32 sums in another order)
    with pytest.raises(native.NativeError):                                                # odd ldb / N of the bf16 k-major operand
        native.check(L.dldkd_gemm_bf16_mixed(1, native.ptr(dy), native.ptr(z16), None, native.ptr(dw), N, K - 1, M, N, K - 1, K - 1, 0, None, 0, None,
                                             native.stream()), "dw")


@pytest.mark.parametrize("M,N,K,relu", [(16384, 384, 384, True), (19200, 1152, 384, False), (1000, 130, 96, True), (2049, 384, 3072, False),
                                        (1024, 64, 32, True)])
def test_gemm_bf16_nt_dma_is_bit_identical_to_gemm_bf16(bf16_mode, M, N, K, relu):
    """dldkd_gemm_bf16_nt (operand tiles HBM -> LDS by LDS-DMA as fp32, rounded to bf16 at the fragment read) against
    dldkd_gemm_bf16 (rounded on the way into the LDS): the same products in the same order - equal bit for bit, ragged row and
    column edges included; ops.linear / the dX layout of ops.gemm route through it in throughput mode."""
    from dldkd_amd import native, ops
    L = native.lib()
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.1).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    assert L.dldkd_gemm_bf16_nt_ok(M, N, K, K, K)
    y_new, y_old = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    native.check(L.dldkd_gemm_bf16_nt(native.ptr(a), native.ptr(w), native.ptr(b), native.ptr(y_new), M, N, K, K, K, N, int(relu), None,
                                      native.stream()), "nt")
    native.check(L.dldkd_gemm_bf16(native.ptr(a), native.ptr(w), native.ptr(b), native.ptr(y_old), M, N, K, K, K, N, 0, 0, int(relu), None, 0,
                                   native.stream()), "old")
    assert torch.equal(y_new, y_old)
    ref = a.to(torch.bfloat16).double() @ w.to(torch.bfloat16).double().T + b.double()
    ref = ref.clamp_min(0) if relu else ref
    assert (y_new.double() - ref).abs().max().item() < 1e-3 * max(1.0, ref.abs().max().item())
    # the wrappers: forward and dX (weight transposed on the fly) against the register-staged kernel
    dy = torch.randn(M, N, generator=g).to(DEV)
    try:
        ops.GEMM_NT_DMA = True
        y1, dx1 = ops.linear(a, w, b, relu=relu), ops.gemm(dy, w, False, True, M, K, N)
        ops.GEMM_NT_DMA = False
        y0, dx0 = ops.linear(a, w, b, relu=relu), ops.gemm(dy, w, False, True, M, K, N)
    finally:
        ops.GEMM_NT_DMA = True
    assert torch.equal(y1, y0) and torch.equal(dx1, dx0)
    assert not L.dldkd_gemm_bf16_nt_ok(M, N, K + 8, K + 8, K + 8)


@pytest.mark.parametrize("nv,L,K,p_drop", [(128, 128, 3072, 0.2), (37, 64, 512, 0.0), (16, 96, 256, 0.3)])
def test_training_input_projection_skips_the_padding(bf16_mode, nv, L, K, p_drop):
    """functional.in_proj_train with the batch's mask: rows of the padding are never read (NaN features there change nothing) and
    come out as relu(bias); dW skips the 32-row groups without a valid row.  Against the same call without the mask on the
    zero-padded batch: valid rows of the output bit-identical, gradients of W / b / gamma / beta equal up to summation order
    when the upstream gradient is zero on the padding (as every loss makes it)."""
    from dldkd_amd import functional as F_
    g = torch.Generator().manual_seed(nv + L + K)
    lens = torch.randint(1, L + 1, (nv,), generator=g)
    lens[0] = L
    mask = (torch.arange(L)[None] < lens[:, None]).float()
    x = torch.randn(nv, L, K, generator=g) * mask[..., None]
    gamma, beta = 1 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    W, b = torch.randn(384, K, generator=g) * 0.02, 0.1 * torch.randn(384, generator=g)
    w_out = (torch.randn(nv, L, 384, generator=g) * mask[..., None]).to(DEV)          # d loss / d y: zero on the padding
    res = []
    for use_mask in (True, False):
        gs, bs, Ws, bbs = (t.to(DEV).requires_grad_() for t in (gamma, beta, W, b))
        xd = x.to(DEV)
        if use_mask:
            xd = torch.where(mask.to(DEV)[..., None] > 0, xd, torch.full_like(xd, float("nan")))      # the padding must not be read
        torch.manual_seed(77)
        y = F_.in_proj_train(xd, gs, bs, Ws, bbs, p_drop, True, row_mask=mask.to(DEV) if use_mask else None)
        (y * w_out).sum().backward()
        res.append((y.detach().cpu(), gs.grad.cpu(), bs.grad.cpu(), Ws.grad.cpu(), bbs.grad.cpu()))
    (ym, dgm, dbm, dWm, dbbm), (y0, dg0, db0, dW0, dbb0) = res
    valid = mask.bool()
    assert torch.equal(ym[valid], y0[valid]) and torch.isfinite(ym).all()
    assert torch.equal(ym[~valid], torch.relu(b).expand(int((~valid).sum()), -1))
    for a, r in ((dWm, dW0), (dbbm, dbb0), (dgm, dg0), (dbm, db0)):
        assert torch.isfinite(a).all() and torch.allclose(a, r, rtol=1e-4, atol=1e-5 * r.abs().max().item())


def test_video_tower_skips_the_padding_without_changing_the_step(bf16_mode):
    """Training, throughput mode: with the batch's mask the input projection flags the 32-row groups that hold valid clips and
    every row-wise kernel of the video towers (LayerNorms, linears, their backward passes) skips the others (ops.row_groups).
    No loss term reads a padded clip and padded keys are masked out of attention, so losses and all 74 gradients must equal
    those of the run that computes the padding too - up to fp32 summation order - with dropout on (masks are indexed by
    position, the same in both runs)."""
    import types
    from dldkd_amd import functional as F_
    from dldkd_amd.model import DLDKD
    cfg = types.SimpleNamespace(visual_input_size=512, query_input_size=256, inheritance_hidden=384, exploration_hidden=384,
                                max_ctx_l=64, max_desc_l=30, input_drop=0.2, drop=0.2, n_heads=4, initializer_range=0.02,
                                margin=0.1, use_hard_negative=False, hard_pool_size=20, label_style="soft")
    mopt = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                 collection="tvr", alpha=0.8, belta=0.8)
    batch = synth.make_train_batch(31, nv=48, caps=3, L=64, len_lo=5, dv=512, dq=256)
    batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    assert batch["student_videos"].shape[1] == 64 and float(batch["student_videos_mask"].mean()) < 0.8
    res = []
    try:
        for skip in (True, False):
            F_.IN_PROJ_SKIP_PADDING = DLDKD.TOWER_SKIPS_PADDING = skip
            torch.manual_seed(5)
            m = DLDKD(types.SimpleNamespace(**vars(cfg)), mopt).to(DEV).train()
            torch.manual_seed(9)
            loss, parts = m(batch)
            loss.backward()
            res.append((float(loss), {k: float(v) for k, v in parts.items() if torch.is_tensor(v)},
                        {n: p.grad.detach().clone() for n, p in m.named_parameters()}))
    finally:
        F_.IN_PROJ_SKIP_PADDING = DLDKD.TOWER_SKIPS_PADDING = True
    (la, pa, ga), (lb, pb, gb) = res
    assert la == pytest.approx(lb, rel=1e-5)
    for k in pa:
        assert pa[k] == pytest.approx(pb[k], rel=1e-4, abs=1e-6), k
    assert len(ga) == 74
    for n in ga:
        assert torch.isfinite(ga[n]).all(), n
        scale = gb[n].abs().max().item()
        assert (ga[n] - gb[n]).abs().max().item() <= 2e-4 * max(scale, 1e-6) + 1e-7, (n, scale)


@pytest.mark.parametrize("nv,L,K,N,p_drop,masked", [(128, 128, 3072, 384, 0.2, True), (40, 30, 768, 384, 0.1, False), (9, 64, 256, 130, 0.0, True),
                                                     (300, 32, 512, 384, 0.5, True)])
def test_input_projection_backward_as_one_two_accumulator_gemm(bf16_mode, nv, L, K, N, p_drop, masked):
    """dldkd_inproj_bwd_bf16 (functional.IN_PROJ_BWD_DUAL): dW and the LayerNorm parameter gradients of the training input
    projection from ONE weight-gradient GEMM with a second accumulator set against the mask [z != 0] - the (M, K) product dy W of
    dldkd_linear_lngrad reassociated.  Against the two-GEMM path it replaces (same saved rows, same dropout mask: dW and the bias
    gradient equal up to summation order, dgamma / dbeta within the bf16 rounding of the saved rows) and against fp64 on the exact
    normalised rows; columns with gamma = 0 or tiny (nothing of xhat left in z) go through the exact per-column path, including
    gamma = beta = 0 where the mask [z != 0] itself is empty."""
    from dldkd_amd import functional as F_
    g = torch.Generator().manual_seed(nv + L + K)
    lens = torch.randint(1, L + 1, (nv,), generator=g)
    lens[0] = L
    mask = (torch.arange(L)[None] < lens[:, None]).float() if masked else torch.ones(nv, L)
    x = (torch.randn(nv, L, K, generator=g) * (1 + torch.rand(nv, L, 1, generator=g)) + 0.3) * mask[..., None]
    gamma, beta = 1 + 0.2 * torch.randn(K, generator=g), 0.2 * torch.randn(K, generator=g)
    gamma[3], gamma[7], gamma[K - 1], gamma[11] = 0.0, 1e-3, -0.02, 0.0
    beta[11] = 0.0
    W, b = torch.randn(N, K, generator=g) * 0.03, 0.1 * torch.randn(N, generator=g)
    w_out = (torch.randn(nv, L, N, generator=g) * mask[..., None]).to(DEV)          # d loss / d y: zero on the padding
    res = {}
    try:
        for dual in (True, False):
            F_.IN_PROJ_BWD_DUAL = dual
            gs, bs, Ws, bbs = (t.to(DEV).requires_grad_() for t in (gamma, beta, W, b))
            torch.manual_seed(5)
            y = F_.in_proj_train(x.to(DEV), gs, bs, Ws, bbs, p_drop, True, row_mask=mask.to(DEV) if masked else None)
            (y * w_out).sum().backward()
            res[dual] = [t.grad.double().cpu() for t in (gs, bs, Ws, bbs)] + [y.detach()]
    finally:
        F_.IN_PROJ_BWD_DUAL = True
    (dg1, db1, dW1, dbb1, y1), (dg0, db0, dW0, dbb0, y0) = res[True], res[False]
    assert torch.equal(y1, y0)
    assert torch.allclose(dW1, dW0, rtol=1e-4, atol=1e-5 * dW0.abs().max().item())
    # the one-GEMM path takes the bias gradient from the bf16 copy of dy its GEMM reads (gemm_bf16_tn.hip: column sums of the A
    # fragments), the two-GEMM path from the fp32 rows on their way to LDS: equal up to the bf16 rounding of 16,384 summands
    assert ((dbb1 - dbb0).norm() / dbb0.norm()).item() <= 3e-3
    rel = lambda a, r: ((a - r).norm() / r.norm()).item()                                   # noqa: E731
    # the two paths round different things (two GEMMs: dy and W to bf16; one GEMM: dy and, for dgamma, the saved rows): a few 1e-3
    print(f"  one GEMM vs two: dgamma rel l2 {rel(dg1, dg0):.2e}, dbeta {rel(db1, db0):.2e}")
    assert rel(dg1, dg0) <= 5e-3 and rel(db1, db0) <= 5e-3
    sg, sb = dg0.abs().max().item(), db0.abs().max().item()
    for k in (3, 7, K - 1, 11):                                                             # the exact columns (fp32 x, W, dy)
        assert abs(dg1[k] - dg0[k]) <= 5e-3 * sg and abs(db1[k] - db0[k]) <= 5e-3 * sb, (k, dg1[k], dg0[k], db1[k], db0[k])
    assert torch.isfinite(dg1).all() and torch.isfinite(db1).all()
    if p_drop == 0.0:
        # no dropout: fp64 from the definition, with the ReLU mask the kernels saw - both paths against it
        xd, gd, bd, Wd = x.double().reshape(-1, K), gamma.double(), beta.double(), W.double()
        mu = xd.mean(1, keepdim=True)
        xhat = (xd - mu) / torch.sqrt(((xd - mu) ** 2).mean(1, keepdim=True) + 1e-5)
        dyr = (w_out.double().cpu() * (y1.double().cpu() > 0)).reshape(-1, N) * mask.double().reshape(-1, 1)
        dz = dyr @ Wd
        ref_g, ref_b = (dz * xhat).sum(0), dz.sum(0)
        print(f"  vs fp64: one GEMM dgamma {rel(dg1, ref_g):.2e} dbeta {rel(db1, ref_b):.2e}; two GEMMs {rel(dg0, ref_g):.2e} {rel(db0, ref_b):.2e}")
        assert rel(dg1, ref_g) <= 5e-3 and rel(db1, ref_b) <= 5e-3
        for k in (3, 7, K - 1, 11):
            assert abs(dg1[k] - ref_g[k]) <= 1e-4 * sg and abs(db1[k] - ref_b[k]) <= 1e-4 * sb


def test_row_group_flags_are_the_or_over_the_32_rows_of_a_group(bf16_mode):
    """The padding skip's group flags (written by the LayerNorm-dropout kernel of the training input projection) say "some row of
    these 32 is valid" for ANY mask - not only for prefix masks, whose first row decides: a scattered mask flags every
    group that holds a valid row, and the rows are normalised exactly where the mask says."""
    from dldkd_amd import native
    L = native.lib()
    g = torch.Generator().manual_seed(3)
    M, K = 32 * 40, 256
    x = torch.randn(M, K, generator=g).to(DEV)
    mask = (torch.rand(M, generator=g) < 0.05).float()
    mask[32 * 7:32 * 8] = 0                                              # an empty group
    mask[32 * 9] = 0; mask[32 * 9 + 31] = 1                              # a group whose first row is padding and last row valid
    gamma, beta = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
    z = torch.empty(M, K, dtype=torch.bfloat16, device=DEV)
    stats = torch.empty(2, M, device=DEV)
    flags = torch.full((M // 32,), 7, dtype=torch.uint8, device=DEV)
    native.check(L.dldkd_layernorm_dropout_bf16(native.ptr(x), native.ptr(gamma), native.ptr(beta), native.ptr(z), None, native.ptr(stats), M, K,
                                                1e-5, 0.0, 0, 0, None, native.ptr(mask.to(DEV)), native.ptr(flags), native.stream()), "ln")
    want = mask.view(-1, 32).amax(1).to(torch.uint8)
    assert torch.equal(flags.cpu(), want) and int(want[7]) == 0 and int(want[9]) == 1
    ref = torch.nn.functional.layer_norm(x.cpu(), (K,), eps=1e-5)
    got = z.float().cpu()
    assert torch.equal(got[mask == 0], torch.zeros_like(got[mask == 0]))
    assert (got[mask > 0] - ref[mask > 0]).abs().max().item() < 2e-2


@pytest.mark.parametrize("nq,nv,L,D", [(640, 128, 128, 384), (257, 128, 64, 384), (70, 5, 33, 128)])
def test_pooled_forward_from_bf16_operands_matches_the_fp32_operand_kernel(nq, nv, L, D):
    """gemm_bf16_nt16_pool_kernel (bf16 rows cast by the norm pass, LDS-DMA tiles, row tiles past a video's length skipped) against
    gemm_bf16_pool_kernel (fp32 rows rounded on their way to LDS): the same rounded operands, so the pooled maxima agree to fp32
    summation order, the arg-max clips agree except at such near-ties, and masked / empty videos give the same constants."""
    from dldkd_amd import functional as F_
    from dldkd_amd import ops
    gen = torch.Generator().manual_seed(nq + L)
    q = torch.randn(nq, D, generator=gen).to(DEV)
    g = torch.randn(nv, L, D, generator=gen).to(DEV)
    lens = torch.randint(1, L + 1, (nv,), generator=gen)
    lens[1] = 0
    lens[2] = L
    labels = torch.randint(0, nv, (nq,), generator=gen)
    labels[labels == 1] = 0
    lens_d, labels_d = lens.to(DEV).int(), labels.to(DEV).int()
    ops.set_gemm_precision("bf16")
    old = F_.SIMPOOL_TRAIN_BF16_OPERANDS
    try:
        F_.SIMPOOL_TRAIN_BF16_OPERANDS = True
        new = F_._SimPoolTrain.apply(q, g, lens_d, labels_d, True)
        F_.SIMPOOL_TRAIN_BF16_OPERANDS = False
        ref = F_._SimPoolTrain.apply(q, g, lens_d, labels_d, True)
    finally:
        F_.SIMPOOL_TRAIN_BF16_OPERANDS = old
        ops.set_gemm_precision("fp32")
    pc, pr, ac, ar, clip = [t.cpu() for t in new]
    pc0, pr0, ac0, ar0, clip0 = [t.cpu() for t in ref]
    scale = float(pr0[:, lens > 0].abs().max())
    assert float((pr - pr0).abs().max()) <= 2e-5 * scale
    assert float((pc - pc0).abs().max()) <= 2e-5
    assert float((clip - clip0).abs().max()) <= 2e-5
    assert bool((pr[:, 1] == -1e10).all()) and bool((pc[:, 1] == -1e10).all())
    assert float((ar != ar0).float().mean()) < 1e-3 and float((ac != ac0).float().mean()) < 1e-3
    assert bool((ar < lens.clamp(min=1)[None]).all()) and bool((ac < lens.clamp(min=1)[None]).all())
