"""GPU parity of the fp32 encoder kernels: each op against a plain torch fp32/fp64 reference of the same
op, the towers against the oracle and against the committed reference outputs (golden G2)."""
import types

import numpy as np
import pytest
import torch

import dldkd_oracle as orc
import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.fixture(params=["fp32", "fp32_exact"])
def fp32_kind(request):
    """both parity-grade GEMMs: the three-plane bf16 kernel (default "fp32") and the true fp32-input MFMA"""
    from dldkd_amd import ops
    ops.set_gemm_precision(request.param)
    yield request.param
    ops.set_gemm_precision("fp32")


@pytest.mark.parametrize("M,N,K", [(1, 1, 4), (130, 70, 20), (257, 384, 3072), (1000, 1152, 384), (64, 384, 768)])
def test_gemm_linear_forward(fp32_kind, M, N, K):
    from dldkd_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05, torch.randn(N, generator=g)
    y = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), relu=True)
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    assert _rel(y, ref) < 2e-6          # fp32-grade products, fp32 accumulation; tolerance = accumulation-order noise
    y2 = ops.linear(x.to(DEV), w.to(DEV))
    assert _rel(y2, x.double() @ w.double().t()) < 2e-6


@pytest.mark.parametrize("M,N,K", [(96, 50, 36), (300, 384, 384), (640, 384, 16384)])
def test_gemm_backward_layouts(fp32_kind, M, N, K):
    """dX = dY.W (b_kmajor) and dW = dY^T.X (both kmajor) without transposes."""
    from dldkd_amd import ops
    g = torch.Generator().manual_seed(7)
    dy, w, x = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g), torch.randn(M, N, generator=g)
    dx = ops.gemm(dy.to(DEV), w.to(DEV), False, True, M, N, K)            # sum_k dy[m,k] w[k,n]
    # one fp32 fmaf chain of length K: rounding noise grows ~ sqrt(K) * 2^-24
    assert _rel(dx, dy.double() @ w.double()) < 2e-6 * max(1.0, (K / 512) ** 0.5)
    Kc = 128 if K > 128 else K
    dw = ops.gemm(dy[:, :Kc].contiguous().to(DEV), x.to(DEV), True, True, Kc, N, M)   # sum_m dy[m,a] x[m,b]
    assert _rel(dw, dy[:, :Kc].double().t() @ x.double()) < 2e-6


@pytest.mark.parametrize("D", [384, 768, 1024, 3072])
def test_layernorm_variants(D):
    from dldkd_amd import ops
    g = torch.Generator().manual_seed(D)
    x = torch.randn(5, 9, D, generator=g) * 2 + 0.7
    gam, bet = torch.randn(D, generator=g), torch.randn(D, generator=g)
    ref = torch.nn.functional.layer_norm(x.double(), (D,), gam.double(), bet.double(), 1e-5)
    assert _rel(ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV)), ref) < 2e-6
    pos = torch.randn(9, D, generator=g)
    ref = torch.nn.functional.layer_norm(x.double() + pos.double(), (D,), gam.double(), bet.double(), 1e-5)
    assert _rel(ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV), add=pos.to(DEV), add_mod=9), ref) < 2e-6
    res = torch.randn(5, 9, D, generator=g)
    ref = torch.nn.functional.layer_norm(x.double() + res.double(), (D,), gam.double(), bet.double(), 1e-5)
    assert _rel(ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV), add=res.to(DEV), add_mod=0), ref) < 2e-6


@pytest.mark.parametrize("N,L", [(3, 1), (2, 12), (4, 30), (3, 33), (2, 96), (3, 128)])
def test_attention_vs_oracle(N, L):
    from dldkd_amd import ops
    g = torch.Generator().manual_seed(L)
    x = torch.randn(N, L, 384, generator=g)
    lens = torch.randint(1, L + 1, (N,), generator=g); lens[0] = L
    mask = (torch.arange(L).unsqueeze(0) < lens.unsqueeze(1)).float()
    p = {f"a.{n}.{k}": (torch.randn(384, 384, generator=g) * 0.05 if k == "weight" else torch.randn(384, generator=g) * 0.1)
         for n in ("query", "key", "value") for k in ("weight", "bias")}
    ref = orc.self_attention(x.double(), mask.double(), {k: v.double() for k, v in p.items()}, "a", 4)
    w = torch.cat([p["a.query.weight"], p["a.key.weight"], p["a.value.weight"]], 0)
    b = torch.cat([p["a.query.bias"], p["a.key.bias"], p["a.value.bias"]], 0)
    qkv = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV))
    out = ops.attention(qkv, mask.to(DEV))
    assert _rel(out, ref) < 5e-6
    out_nomask = ops.attention(qkv, None)
    assert _rel(out_nomask, orc.self_attention(x.double(), None, {k: v.double() for k, v in p.items()}, "a", 4)) < 5e-6


def test_modpool_vs_oracle():
    from dldkd_amd import ops
    g = torch.Generator().manual_seed(3)
    h = torch.randn(7, 30, 384, generator=g)
    lens = torch.tensor([30, 1, 5, 17, 29, 2, 11])
    mask = (torch.arange(30).unsqueeze(0) < lens.unsqueeze(1)).float()
    w = torch.randn(1, 384, generator=g) * 0.2
    ref = orc.modular_pool(h.double(), mask.double(), w.double())
    out, attn = ops.modpool(h.to(DEV), mask.to(DEV), w.reshape(-1).to(DEV), want_attn=True)
    assert _rel(out, ref) < 5e-6
    assert abs(attn.sum(1).cpu() - 1).max() < 1e-5 and (attn.cpu()[mask == 0] == 0).all()


def _model(dv, dq, params):
    from dldkd_amd.model import DLDKD
    cfg = types.SimpleNamespace(visual_input_size=dv, query_input_size=dq, inheritance_hidden=384, exploration_hidden=384,
                                max_ctx_l=128, max_desc_l=30, input_drop=0.2, drop=0.2, n_heads=4, initializer_range=0.02,
                                margin=0.1, use_hard_negative=False, hard_pool_size=20, label_style="soft")
    opt = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                collection="tvr", alpha=0.8, belta=0.8)
    m = DLDKD(cfg, opt)
    m.load_state_dict(params, strict=True)
    return m.to(DEV).eval()


def test_state_dict_keys_match_reference():
    m = _model(3072, 768, synth.make_params(1, 3072, 768))
    assert list(m.state_dict().keys()) == list(synth.param_shapes(3072, 768).keys())
    assert sum(p.numel() for p in m.parameters()) == 5755392      # SURVEY: TVR model size


@pytest.mark.parametrize("tag,dv,dq,seed", [("tvr", 3072, 768, 21), ("anet", 1024, 1024, 22)])
def test_towers_vs_golden_g2(golden_dir, tag, dv, dq, seed):
    g = np.load(f"{golden_dir}/g2_encoders.npz")
    m = _model(dv, dq, synth.make_params(seed, dv, dq))
    rs = np.random.RandomState(seed + 100)
    vid, vmask = synth.make_videos(rs, 6, 12, dv, g[f"{tag}_vlens"])
    txt, tmask = synth.make_texts(rs, 5, 30, dq, g[f"{tag}_qlens"])
    vid, vmask, txt, tmask = [torch.from_numpy(a.astype(np.float32)).to(DEV) for a in (vid, vmask, txt, tmask)]
    with torch.no_grad():
        gi, ge = m.encode_context(vid, vmask)
        qi, qe = m.encode_query(txt, tmask)
    for out, key in ((gi, "ctx_inh"), (ge, "ctx_exp"), (qi, "q_inh"), (qe, "q_exp")):
        ref = g[f"{tag}_{key}"]
        err = np.abs(out.cpu().numpy() - ref).max()
        assert err <= 2e-5 * max(1.0, np.abs(ref).max()), (key, err)
