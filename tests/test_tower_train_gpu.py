"""GPU: the fused training towers (tower_train.hip + the bf16-io attention kernels, functional._TowerTrain) - everything of an encoder
tower behind its input projection as 2 + 2 row kernels - against an fp64 restatement of the reference layers
(method/model_components.py:277-284, 398-450; method/model.py:219) and against the unfused kernel chain it replaces (same Philox
dropout masks, bit for bit)."""
import types

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H = 384


def _ref_tower(y0, pos, g1, b1, wq, bq, wk, bk, wv, bv, wd, bd, g2, b2, wo, bo, mask):
    """fp64: TrainablePositionalEncoding -> BertSelfAttention -> BertSelfOutput [-> out_mapping_linear], dropout off."""
    N, L, _ = y0.shape
    ln = lambda x, g, b: torch.nn.functional.layer_norm(x, (H,), g, b, 1e-5)      # noqa: E731
    h1 = ln(y0 + pos[:L].unsqueeze(0), g1, b1)
    q, k, v = h1 @ wq.t() + bq, h1 @ wk.t() + bk, h1 @ wv.t() + bv
    sp = lambda x: x.view(N, L, 4, 96).permute(0, 2, 1, 3)                          # noqa: E731
    s = sp(q) @ sp(k).transpose(-1, -2) / 96 ** 0.5 + (1.0 - mask)[:, None, None, :] * -10000.0
    ctx = (torch.softmax(s, -1) @ sp(v)).permute(0, 2, 1, 3).reshape(N, L, H)
    h2 = ln(ctx @ wd.t() + bd + h1, g2, b2)
    return h2 if wo is None else h2 @ wo.t() + bo


def _params(rs, video):
    w = lambda: torch.from_numpy(rs.standard_normal((H, H)) * 0.05)                 # noqa: E731
    b = lambda: torch.from_numpy(rs.standard_normal(H) * 0.05)                      # noqa: E731
    g = lambda: torch.from_numpy(1.0 + 0.1 * rs.standard_normal(H))                 # noqa: E731
    p = dict(pos=torch.from_numpy(rs.standard_normal((128, H)) * 0.05), g1=g(), b1=b(), wq=w(), bq=b(), wk=w(), bk=b(), wv=w(), bv=b(),
             wd=w(), bd=b(), g2=g(), b2=b())
    p.update(wo=w() if video else None, bo=b() if video else None)
    return p


ORDER = ("pos", "g1", "b1", "wq", "bq", "wk", "bk", "wv", "bv", "wd", "bd", "g2", "b2", "wo", "bo")


@pytest.mark.parametrize("video,N,L,len_lo", [(True, 24, 128, 20), (True, 10, 64, 3), (True, 7, 50, 5), (False, 130, 30, 4),
                                                (False, 33, 16, 1), (True, 3, 32, 32)])
def test_fused_training_tower_vs_fp64(video, N, L, len_lo):
    """Output and EVERY gradient (input, position rows, both LayerNorms, five weight matrices, five biases) against fp64 autograd;
    the rows of all-padding 32-row groups are poisoned with NaN on the way in (they must never be read) and must come back as
    exact zeros in the input gradient."""
    from dldkd_amd import functional as F_, ops
    rs = np.random.RandomState(1000 + N + L)
    lens = rs.randint(len_lo, L + 1, size=N)
    lens[0] = L
    mask = torch.from_numpy((np.arange(L)[None] < lens[:, None]).astype(np.float64))
    y0 = torch.relu(torch.from_numpy(rs.standard_normal((N, L, H))))
    dout = torch.from_numpy(rs.standard_normal((N, L, H))) * mask.unsqueeze(-1)          # nothing reads a padded clip
    p64 = _params(rs, video)
    ref_in = [y0.clone().requires_grad_(True)] + [None if p64[k] is None else p64[k].clone().requires_grad_(True) for k in ORDER]
    ref = _ref_tower(ref_in[0], *ref_in[1:], mask)
    ref.backward(dout)

    use_flags = L % 32 == 0
    flags = None
    y0d = y0.float().to(DEV)
    if use_flags:
        gv = (np.arange(0, L, 32)[None] < lens[:, None])                                    # group holds a valid clip
        flags = torch.from_numpy(gv.reshape(-1).astype(np.uint8)).to(DEV)
        rowv = torch.from_numpy(np.repeat(gv, 32, axis=1)).to(DEV)
        y0d = torch.where(rowv.unsqueeze(-1), y0d, torch.full_like(y0d, float("nan")))
    y0d.requires_grad_(True)
    pd = {k: (None if v is None else v.float().to(DEV).requires_grad_(True)) for k, v in p64.items()}
    lin = lambda w, b: types.SimpleNamespace(weight=pd[w], bias=pd[b])                     # noqa: E731
    ops.set_gemm_precision("bf16")
    try:
        out = F_.tower_train(y0d, pd["pos"][:L], pd["g1"], pd["b1"], (lin("wq", "bq"), lin("wk", "bk"), lin("wv", "bv")), lin("wd", "bd"),
                             pd["g2"], pd["b2"], lin("wo", "bo") if video else None, mask.float().to(DEV),
                             torch.from_numpy(lens.astype(np.int32)).to(DEV), flags, 0.0, 0.0, 0.0, True, relu_mask=False)
        out.backward(dout.float().to(DEV))
    finally:
        ops.set_gemm_precision("fp32")
    m3 = mask.bool().unsqueeze(-1).expand(N, L, H)
    o = out.detach().double().cpu()
    assert torch.isfinite(o[m3]).all()
    scale = float(ref.detach()[m3].abs().max())
    err = float((o - ref.detach())[m3].abs().max())
    assert err <= 3e-2 * scale, (err, scale)                                               # bf16 operands, K = 384 (measured ~8e-3)
    worst = {}
    got = [y0d.grad] + [None if pd[k] is None else pd[k].grad for k in ORDER]
    bias_scale = max(float(r.grad.norm()) for n_, r in zip(("y0",) + ORDER, ref_in) if r is not None and n_ in ("bq", "bv", "bd"))
    for name, g, r in zip(("y0",) + ORDER, got, ref_in):
        if r is None:
            continue
        g, rg = g.double().cpu(), r.grad
        if name == "pos":
            assert float(g[L:].abs().sum()) == 0.0
        if name == "y0" and use_flags:
            assert float(g[~rowv.cpu()].abs().sum()) == 0.0                      # skipped groups: exact zero rows
        assert torch.isfinite(g).all(), name
        # (the key bias has an exactly-zero gradient - softmax is shift invariant - so its error is measured against the other biases')
        rel = float((g - rg).norm() / (rg.norm().clamp_min(1e-30) if name != "bk" else bias_scale))
        worst[name] = rel
        assert rel <= 4e-2, (name, rel)                                                    # measured <= 1.5e-2
    print("  tower", "video" if video else "query", N, L, "out err", f"{err / scale:.2e}", "worst grad",
          max(worst.items(), key=lambda kv: kv[1]))


def _model(drop):
    from dldkd_amd.model import DLDKD
    cfg = types.SimpleNamespace(visual_input_size=1024, query_input_size=768, inheritance_hidden=384, exploration_hidden=384,
                                max_ctx_l=128, max_desc_l=30, input_drop=drop, drop=drop, n_heads=4, initializer_range=0.02,
                                margin=0.1, use_hard_negative=True, hard_pool_size=20, label_style="soft")
    opt = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                collection="tvr", alpha=0.8, belta=0.8)
    m = DLDKD(cfg, opt)
    m.load_state_dict(synth.make_params(77, 1024, 768), strict=True)
    return m.to(DEV).train()


@pytest.mark.parametrize("drop", [0.0, 0.2])
def test_fused_towers_match_the_unfused_chain_with_the_same_dropout_masks(drop):
    """The whole training forward + backward in throughput mode with the fused towers against the kernel chain they replace, same
    torch seed: the Philox slots are drawn in the same order with the same sizes, so both runs drop the same elements and differ only
    by where activations are rounded to bf16 (the chain keeps fp32 rows between its GEMMs)."""
    from dldkd_amd import functional as F_, ops
    batch = synth.make_train_batch(9, nv=40, caps=3, L=64, len_lo=5, dv=1024, dq=768)
    batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    res = {}
    ops.set_gemm_precision("bf16")
    try:
        for fused in (True, False):
            F_.TOWER_TRAIN_FUSED = fused
            m = _model(drop)
            torch.manual_seed(5)
            loss, d = m(batch)
            m.zero_grad()
            loss.backward()
            res[fused] = (float(loss), {k: float(v) for k, v in d.items()}, {n: p.grad.detach().double().cpu() for n, p in m.named_parameters()})
    finally:
        F_.TOWER_TRAIN_FUSED = True
        ops.set_gemm_precision("fp32")
    (lf, df, gf), (lu, du, gu) = res[True], res[False]
    assert abs(lf - lu) <= 5e-3 * abs(lu), (lf, lu)
    for k in df:
        assert abs(df[k] - du[k]) <= 1e-2 * max(abs(du[k]), 0.05), (k, df[k], du[k])
    nmax = max(float(g.norm()) for g in gu.values())
    worst = 0.0
    for n in gu:
        rel = float((gf[n] - gu[n]).norm()) / max(float(gu[n].norm()), 1e-3 * nmax)
        worst = max(worst, rel)
        assert rel <= 0.15, (n, rel)                        # two bf16-grade evaluations of the same function
    print(f"  fused vs unfused (drop {drop}): loss {lf:.5f} / {lu:.5f}, worst gradient rel l2 difference {worst:.3e}")


@pytest.mark.parametrize("M,N,K,relu,flags", [(2048, 384, 3072, True, True), (1280, 384, 768, False, False), (130, 200, 64, True, False),
                                                (1024, 1152, 1024, False, True)])
def test_gemm_bf16_nt16_vs_fp64(M, N, K, relu, flags):
    """The bf16 x bf16 LDS-DMA GEMM of the training input projection (gemm_bf16_dma.hip): exact products of the bf16 operands with
    fp32 accumulation against fp64, bias / ReLU epilogue, ragged M / N, and the 32-row group filter (rows of skipped groups come out
    as act(bias))."""
    from dldkd_amd import native
    L_ = native.lib()
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05)
    bias = torch.randn(N, generator=g)
    w16 = torch.empty(N, K, dtype=torch.bfloat16, device=DEV)
    native.check(L_.dldkd_cast_bf16(native.ptr(w.to(DEV)), native.ptr(w16), N * K, native.stream()), "cast")
    assert torch.equal(w16.cpu(), w.to(torch.bfloat16))                                  # round to nearest even, like torch
    fl = None
    keep = torch.ones(M, dtype=torch.bool)
    if flags:
        fb = (torch.rand(M // 32, generator=g) > 0.3)
        fl = fb.to(torch.uint8).to(DEV)
        keep = fb.repeat_interleave(32)
    y = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ad = a.to(DEV)
    assert L_.dldkd_gemm_bf16_nt16_ok(M, N, K, K, K)
    native.check(L_.dldkd_gemm_bf16_nt16(native.ptr(ad), native.ptr(w16), native.ptr(bias.to(DEV)), native.ptr(y), M, N, K, K, K, N, int(relu),
                                         native.ptr(fl), native.stream()), "nt16")
    ref = a.double() @ w.to(torch.bfloat16).double().t() + bias.double()
    ref[~keep] = bias.double()
    if relu:
        ref = ref.clamp_min(0)
    err = (y.cpu().double() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err


def test_backward_without_position_gradient_matches_the_one_with_it():
    """pos.requires_grad = False takes the finishing launch without the position sums (dx1 / dpos NULL): every other gradient is
    the one the full backward pass gives (bit for bit where no atomics are involved: the LayerNorm and bias sums end in fp32 atomics)."""
    from dldkd_amd import functional as F_, ops
    rs = np.random.RandomState(5)
    N, L = 6, 64
    lens = rs.randint(10, L + 1, size=N)
    mask = torch.from_numpy((np.arange(L)[None] < lens[:, None]).astype(np.float32)).to(DEV)
    y0 = torch.relu(torch.from_numpy(rs.standard_normal((N, L, H)).astype(np.float32))).to(DEV)
    dout = torch.from_numpy(rs.standard_normal((N, L, H)).astype(np.float32)).to(DEV) * mask.unsqueeze(-1)
    p64 = _params(rs, True)
    res = {}
    ops.set_gemm_precision("bf16")
    try:
        for need_pos in (True, False):
            pd = {k: v.float().to(DEV).requires_grad_(need_pos or k != "pos") for k, v in p64.items()}
            lin = lambda w, b: types.SimpleNamespace(weight=pd[w], bias=pd[b])             # noqa: E731
            yd = y0.clone().requires_grad_(True)
            out = F_.tower_train(yd, pd["pos"][:L], pd["g1"], pd["b1"], (lin("wq", "bq"), lin("wk", "bk"), lin("wv", "bv")), lin("wd", "bd"),
                                 pd["g2"], pd["b2"], lin("wo", "bo"), mask, torch.from_numpy(lens.astype(np.int32)).to(DEV), None,
                                 0.0, 0.0, 0.0, True, relu_mask=False)
            out.backward(dout)
            res[need_pos] = {k: pd[k].grad for k in ORDER if k != "pos"}
            res[need_pos]["y0"] = yd.grad
            assert (pd["pos"].grad is not None) == need_pos
    finally:
        ops.set_gemm_precision("fp32")
    for k, g in res[True].items():
        r = res[False][k]
        if k in ("g1", "b1", "g2", "b2", "bq", "bk", "bv", "bd", "bo"):            # sums that end in fp32 atomics
            assert torch.allclose(g, r, rtol=1e-4, atol=1e-5 * float(g.abs().max())), k
        else:
            assert torch.equal(g, r), k
