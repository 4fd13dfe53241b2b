"""GPU: both video towers' input-projection LayerNorm in ONE pass over the raw features (dldkd_layernorm_dropout_bf16_dual,
functional.in_proj_ln_dual; reference: LinearLayer.forward's LayerNorm -> Dropout once per branch on the SAME student features,
method/model.py:229-243, model_components.py:305-310).  The dual launch must write exactly what two single launches write, and a
training step through it must be the step without it."""
import types

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("nv,L,K,p", [(9, 64, 3072, 0.2), (5, 32, 1024, 0.0), (3, 128, 3072, 0.5), (4, 96, 512, 0.15)])
def test_dual_launch_writes_what_two_single_launches_write(nv, L, K, p):
    from dldkd_amd import native, ops
    lib = native.lib()
    g = torch.Generator(device=DEV).manual_seed(nv * 7 + L)
    M = nv * L
    x = torch.randn(M, K, generator=g, device=DEV) * 3 + 0.5
    lens = torch.randint(1, L + 1, (nv,), generator=g, device=DEV)
    lens[0] = L
    mask = (torch.arange(L, device=DEV)[None, :] < lens[:, None]).float().reshape(-1).contiguous()
    gam = [torch.randn(K, generator=g, device=DEV) for _ in range(2)]
    bet = [torch.randn(K, generator=g, device=DEV) for _ in range(2)]
    seed, offs = 1234567, (4096, 4096 + 4 * ((M * K + 3) // 4))
    z = [torch.full((M, K), 7.0, dtype=torch.bfloat16, device=DEV) for _ in range(2)]
    stats = torch.empty(2, M, device=DEV)
    gfl = torch.empty(M // 32, dtype=torch.uint8, device=DEV)
    P = native.ptr
    native.check(lib.dldkd_layernorm_dropout_bf16_dual(P(x), P(gam[0]), P(bet[0]), P(gam[1]), P(bet[1]), P(z[0]), P(z[1]), P(stats), M, K,
                                                      ops.LN_EPS, p, seed, offs[0], offs[1], None, P(mask), P(gfl), 0, native.stream()), "dual")
    for b in range(2):
        zr = torch.full((M, K), 9.0, dtype=torch.bfloat16, device=DEV)
        sr = torch.empty(2, M, device=DEV)
        gr = torch.empty(M // 32, dtype=torch.uint8, device=DEV)
        native.check(lib.dldkd_layernorm_dropout_bf16(P(x), P(gam[b]), P(bet[b]), P(zr), None, P(sr), M, K, ops.LN_EPS, p, seed, offs[b],
                                                     None, P(mask), P(gr), native.stream()), "single")
        assert torch.equal(z[b].view(torch.int16), zr.view(torch.int16)), b
        assert torch.equal(stats, sr) and torch.equal(gfl, gr)
    if p > 0:
        assert not torch.equal(z[0] == 0, z[1] == 0)                   # the branches draw their own masks
    # the device-state form (hipGraph replays): base offset added to both
    st = torch.tensor([seed, 1000], dtype=torch.int64, device=DEV)
    z2 = [torch.empty_like(z[0]) for _ in range(2)]
    native.check(lib.dldkd_layernorm_dropout_bf16_dual(P(x), P(gam[0]), P(bet[0]), P(gam[1]), P(bet[1]), P(z2[0]), P(z2[1]), P(stats), M, K,
                                                      ops.LN_EPS, p, 0, offs[0] - 1000, offs[1] - 1000, P(st), P(mask), P(gfl), 0,
                                                      native.stream()), "dual_state")
    assert torch.equal(z2[0].view(torch.int16), z[0].view(torch.int16)) and torch.equal(z2[1].view(torch.int16), z[1].view(torch.int16))
    # the two-plane form ("mixed" precision): [2][M][K] per branch = what dldkd_layernorm_ex_f32 writes as out_planes; plane 0 = the bf16 rows
    zp = [torch.empty(2, M, K, dtype=torch.bfloat16, device=DEV) for _ in range(2)]
    native.check(lib.dldkd_layernorm_dropout_bf16_dual(P(x), P(gam[0]), P(bet[0]), P(gam[1]), P(bet[1]), P(zp[0]), P(zp[1]), P(stats), M, K,
                                                      ops.LN_EPS, p, seed, offs[0], offs[1], None, P(mask), P(gfl), 1, native.stream()), "dual_planes")
    for b in range(2):
        ref = torch.empty(2, M, K, dtype=torch.bfloat16, device=DEV)
        native.check(lib.dldkd_layernorm_ex_f32(P(x), None, 0, P(gam[b]), P(bet[b]), None, None, P(ref), None, P(stats), M, K, ops.LN_EPS, p, seed,
                                                offs[b], None, P(mask), P(gfl), None, native.stream()), "ln_ex_planes")
        assert torch.equal(zp[b].view(torch.int16), ref.view(torch.int16)) and torch.equal(zp[b][0].view(torch.int16), z[b].view(torch.int16))


def _model(drop):
    from dldkd_amd.model import DLDKD
    cfg = types.SimpleNamespace(visual_input_size=3072, query_input_size=768, inheritance_hidden=384, exploration_hidden=384,
                                max_ctx_l=128, max_desc_l=30, input_drop=drop, drop=drop, n_heads=4, initializer_range=0.02,
                                margin=0.1, use_hard_negative=True, hard_pool_size=20, label_style="soft")
    opt = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                collection="tvr", alpha=0.8, belta=0.8)
    m = DLDKD(cfg, opt)
    m.load_state_dict(synth.make_params(43, 3072, 768), strict=True)
    return m.to(DEV).train()


def test_training_step_through_the_dual_launch_is_the_step_without_it(monkeypatch):
    """Dropout 0 (with dropout the two forms draw their Philox slots in a different order - different, equally valid masks): the
    seven losses and all gradients of one bf16-mode step with the one-pass LayerNorm equal those with one launch per branch."""
    from dldkd_amd import functional as F_, native, ops
    batch = synth.make_train_batch(3, nv=32, caps=5, L=64, len_lo=24, dv=3072, dq=768)
    dbatch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    calls = {"dual": 0, "single": 0}
    lib = native.lib()
    real_dual, real_single = lib.dldkd_layernorm_dropout_bf16_dual, lib.dldkd_layernorm_dropout_bf16
    res = {}
    ops.set_gemm_precision("bf16")
    try:
        for on in (True, False):
            monkeypatch.setattr(F_, "IN_PROJ_LN_DUAL", on)
            m = _model(0.0)
            m.tower_streams = False
            m.weight = 1.0
            n_dual = [0]
            orig = F_.in_proj_ln_dual
            monkeypatch.setattr(F_, "in_proj_ln_dual", lambda *a, **k: (n_dual.__setitem__(0, n_dual[0] + 1), orig(*a, **k))[1])
            torch.manual_seed(7)
            loss, d = m(dbatch)
            m.zero_grad()
            loss.backward()
            torch.cuda.synchronize()
            monkeypatch.setattr(F_, "in_proj_ln_dual", orig)
            assert n_dual[0] == (1 if on else 0)
            assert not F_._PRE_LN                                             # both stashed rows were consumed (or dropped)
            res[on] = (float(loss), {k: float(v) for k, v in d.items()}, {n: p.grad.detach().clone() for n, p in m.named_parameters()})
    finally:
        ops.set_gemm_precision("fp32")
    assert res[True][1] == res[False][1] and res[True][0] == res[False][0]
    for n, g in res[True][2].items():
        r = res[False][2][n]
        assert float((g - r).abs().max()) <= 1e-5 * max(float(r.abs().max()), 1e-6), n      # (split-K planes are summed by atomics)


def test_replayed_step_with_dropout_cuts_a_graph_behind_the_dual_launch():
    """The stepper's capture: the one-pass LayerNorm is its own first graph ("pre0"), the video towers' graphs wait for it, and the
    replayed step equals the eager step on the same state (GraphedTrainStep's self-check compares loss and parameters)."""
    from dldkd_amd import ops
    from dldkd_amd import train as T
    batch = synth.make_train_batch(3, nv=32, caps=5, L=64, len_lo=24, dv=3072, dq=768)
    dbatch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    ops.set_gemm_precision("bf16")
    try:
        m = _model(0.2)
        topt = types.SimpleNamespace(grad_clip=-1, lr=3e-4, wd=0.01, lr_warmup_proportion=0.01, n_epoch=5)
        optim = T.make_optimizer(m, topt, 10)
        st = T.GraphedTrainStep(m, optim, topt)
        losses = []
        for i in range(4):
            torch.manual_seed(100 + i)
            loss, d = st(dbatch)
            losses.append(float(loss))
        torch.cuda.synchronize()
        e = next(iter(st.graphs.values()))
        assert st.captures == 1 and st.replays == 3 and getattr(e, "par", None) and e.par.get("pre0") is not None
        assert all(np.isfinite(losses)) and losses[-1] < losses[0] * 1.5
    finally:
        ops.set_gemm_precision("fp32")
