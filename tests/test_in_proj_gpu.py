"""GPU: K4 / K4b input projection of the throughput mode (LayerNorm folded, both branches in one pass, 16-bit MFMA operands in
IEEE fp16 = "h16": csrc/common.hpp) against the oracle and the fp32 parity path, and its effect on end-to-end R@K."""
import numpy as np
import pytest
import torch

import dldkd_oracle as orc
import synth
from test_encoder_gpu import _model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("kernel", ["rows128", "full"])
@pytest.mark.parametrize("dv,M", [(3072, 300), (1024, 129), (768, 1), (64, 515), (128, 1000)])
def test_in_proj_vs_oracle(dv, M, kernel, monkeypatch):
    from dldkd_amd import ops
    monkeypatch.setattr(ops, "INPROJ_KERNEL", kernel)          # dv = 64: rows128 is not applicable and falls back
    m = _model(dv, dv, synth.make_params(7, dv, dv))
    g = torch.Generator().manual_seed(dv + M)
    x = torch.nn.functional.normalize(torch.randn(M, dv, generator=g).abs() + 0.1 * torch.randn(M, dv, generator=g), dim=-1)  # i3d-like: positive mean
    p = {k: v.cpu() for k, v in m.state_dict().items()}
    folded = ops.FoldedInProj([m.visual_input_proj, m.exp_visual_input_proj])
    ys = ops.in_proj_h16(x.to(DEV), folded)
    for y, pre in zip(ys, ("", "exp_")):
        ref = orc.input_projection(x.double(), {k: v.double() for k, v in p.items()}, pre + "visual_input_proj")
        err = (y.double().cpu() - ref).abs().max().item()
        assert err <= 4e-3 * max(1.0, ref.abs().max().item()), (pre, err, ref.abs().max().item())     # fp16 operands, K up to 3072 (bf16: 2.5e-2)
        rel = ((y.double().cpu() - ref).norm() / ref.norm()).item()
        assert rel < 1e-3, rel                                                                          # (bf16 operands: 6e-3)
    # weights are re-folded when a parameter changes
    with torch.no_grad():
        m.visual_input_proj.net[1].bias.add_(1.0)
    y2 = ops.in_proj_h16(x.to(DEV), folded)[0]
    assert (y2 - ys[0]).abs().max() > 0.5


@pytest.mark.parametrize("K,M", [(3072, 128 * 300 + 77), (768, 30001), (128, 129), (3072, 1), (1024, 127), (192, 128 * 600 + 3), (320, 40000)])
def test_rows128_matches_full_kernel(K, M, monkeypatch):
    """The two K4 kernels share one contract: same folded weights, same output up to fp32 summation order (the rows128
    kernel adds the k-tiles in a rotated order per workgroup).  Ragged last tile, one-row input, smallest K."""
    from dldkd_amd import ops, native
    m = _model(K, K, synth.make_params(11, K, K))
    g = torch.Generator().manual_seed(K + M)
    x = torch.nn.functional.normalize(torch.randn(M, K, generator=g).abs() + 0.1 * torch.randn(M, K, generator=g), dim=-1).to(DEV)
    folded = ops.FoldedInProj([m.visual_input_proj, m.exp_visual_input_proj])
    assert native.lib().dldkd_in_proj_h16_rows128_ok(K) == 1
    out = {}
    for kern in ("full", "rows128"):
        monkeypatch.setattr(ops, "INPROJ_KERNEL", kern)
        out[kern] = [y.clone() for y in ops.in_proj_h16(x, folded, relu=(K != 1024))]
    for a, b in zip(out["full"], out["rows128"]):
        assert torch.isfinite(b).all()
        assert (a - b).abs().max().item() <= 4e-5 * max(1.0, a.abs().max().item())
    # rows past M are never written: a guard band behind the outputs stays untouched
    L = native.lib()
    f = folded.get()
    ys = [torch.full((M + 128, 384), -7.0, device=DEV) for _ in range(2)]
    native.check(L.dldkd_in_proj_h16_rows128(native.ptr(x), native.ptr(f.Wf), native.ptr(f.cs), native.ptr(f.bb), native.ptr(ys[0]),
                                              native.ptr(ys[1]), M, K, 1e-5, 1, native.stream()), "rows128")
    for y in ys:
        assert (y[M:] == -7.0).all()


def test_rows128_rejects_unsupported_k():
    from dldkd_amd import native
    L = native.lib()
    assert L.dldkd_in_proj_h16_rows128_ok(64) == 0 and L.dldkd_in_proj_h16_rows128_ok(96) == 0 and L.dldkd_in_proj_h16_rows128_ok(192) == 1
    x = torch.zeros(4, 96, device=DEV)
    y = torch.zeros(4, 384, device=DEV)
    rc = L.dldkd_in_proj_h16_rows128(native.ptr(x), native.ptr(x), native.ptr(x), native.ptr(x), native.ptr(y), native.ptr(y), 4, 96,
                                      1e-5, 1, native.stream())
    assert rc != 0


@pytest.mark.parametrize("K,M", [(3072, 300), (768, 1), (64, 129), (1024, 128 * 3 + 5), (4096, 77), (96, 700), (160, 131)])
def test_parity_grade_in_proj_x3(K, M):
    """in_proj_rows128x3_kernel (parity mode, inference): against fp64 math at the accuracy of the LayerNorm + gemm_f32x3 path
    it replaces, and against that path; the row statistics equal the LayerNorm kernel's bit for bit."""
    from dldkd_amd import ops, native
    from dldkd_amd import functional as F_
    m = _model(K, K, synth.make_params(13, K, K))
    with torch.no_grad():                                   # non-trivial gamma / beta
        for l in (m.visual_input_proj, m.exp_visual_input_proj):
            g = torch.Generator().manual_seed(K)
            l.LayerNorm.weight.add_((0.2 * torch.randn(K, generator=g)).to(DEV))
            l.LayerNorm.bias.add_((0.2 * torch.randn(K, generator=g)).to(DEV))
    g = torch.Generator().manual_seed(K + M)
    x = torch.nn.functional.normalize(torch.randn(M, K, generator=g).abs() + 0.1 * torch.randn(M, K, generator=g), dim=-1).to(DEV)
    assert ops.in_proj_x3_ok(K)
    folded = ops.FoldedInProjX3([m.visual_input_proj, m.exp_visual_input_proj])
    with torch.no_grad():
        ys = ops.in_proj_x3(x, folded)
        for y, l in zip(ys, (m.visual_input_proj, m.exp_visual_input_proj)):
            old = l(x)                                      # LayerNorm kernel + gemm_f32x3 (+ bias, ReLU)
            xn = torch.nn.functional.layer_norm(x.double(), (K,), l.LayerNorm.weight.double(), l.LayerNorm.bias.double(), 1e-5)
            ref = torch.relu(xn @ l.net[1].weight.double().t() + l.net[1].bias.double())
            scale = max(1.0, ref.abs().max().item())
            assert torch.isfinite(y).all()
            assert (y.double() - ref).abs().max().item() <= 6e-6 * scale, ((y.double() - ref).abs().max().item(), scale)
            assert (y - old).abs().max().item() <= 8e-6 * scale
    # row statistics: the LayerNorm kernel's own numbers (LayerNorm with gamma = 1, beta = 0 reproduces (x - mean) * rstd)
    L = native.lib()
    st = torch.empty(2, M, device=DEV)
    native.check(L.dldkd_row_meanrstd_f32(native.ptr(x), native.ptr(st[0]), native.ptr(st[1]), M, K, 1e-5, native.stream()), "stats")
    xhat = F_.layernorm(x, torch.ones(K, device=DEV), torch.zeros(K, device=DEV))
    assert torch.equal(xhat, (x - st[0][:, None]) * st[1][:, None])
    with torch.no_grad():
        # batch-invariant bit for bit: a row's result does not depend on its position in the batch
        if M > 130:
            sub = ops.in_proj_x3(x[129:].contiguous(), folded)
            assert torch.equal(sub[0], ys[0][129:]) and torch.equal(sub[1], ys[1][129:])
        # a parameter change re-folds the planes
        m.visual_input_proj.net[1].bias.add_(1.0)
        assert (ops.in_proj_x3(x, folded)[0] - ys[0]).abs().max() > 0.5


@pytest.mark.parametrize("n_lin,K,M", [(1, 384, 300), (2, 384, 129), (3, 384, 128 * 2 + 1), (1, 64, 5), (3, 1024, 77)])
def test_linear_rows_x3_vs_fp64(n_lin, K, M):
    """The fp32-grade full-row linear (q | k | v side by side, dense, out mapping in parity inference) against fp64 and against
    the gemm_f32x3 path; batch-invariant."""
    from dldkd_amd import ops
    from dldkd_amd import functional as F_
    g = torch.Generator().manual_seed(n_lin * 1000 + K + M)
    lins = [torch.nn.Linear(K, 384).to(DEV) for _ in range(n_lin)]
    with torch.no_grad():
        for l in lins:
            l.weight.copy_(torch.randn(384, K, generator=g) * K ** -0.5)
            l.bias.copy_(torch.randn(384, generator=g) * 0.1)
    x = torch.randn(M, K, generator=g).to(DEV)
    packed = ops.PackedLinearX3(lins)
    with torch.no_grad():
        assert ops.rows_x3_ok(x)
        for relu in (False, True):
            y = ops.linear_rows_x3(x, packed, relu=relu)
            assert y.shape == (M, 384 * n_lin)
            for i, l in enumerate(lins):
                ref = x.double() @ l.weight.double().t() + l.bias.double()
                old = F_.linear(x, l.weight, l.bias, relu=relu)
                if relu:
                    ref = torch.relu(ref)
                part = y[:, 384 * i:384 * (i + 1)]
                scale = max(1.0, ref.abs().max().item())
                assert (part.double() - ref).abs().max().item() <= 3e-6 * scale
                assert (part - old).abs().max().item() <= 4e-6 * scale
        if M > 130:
            assert torch.equal(ops.linear_rows_x3(x[129:].contiguous(), packed), ops.linear_rows_x3(x, packed)[129:])
    with torch.enable_grad():
        assert not ops.rows_x3_ok(x)


def test_in_proj_kernels_past_2_pow_32_elements():
    """The whole C2 gallery in ONE call: 21,793 x 128 rows x 3072 fp32 = 8.6e9 elements (34 GB): row offsets past 2^31 and
    2^32 elements must not wrap.  Slices around those offsets equal a call on the slice alone (bit for bit for the
    batch-invariant parity kernel, to summation order for the bf16 one)."""
    from dldkd_amd import ops
    if torch.cuda.get_device_properties(0).total_memory < 80e9:
        pytest.skip("needs 45 GB of device memory")
    K, M = 3072, 21793 * 128
    m = _model(K, 768, synth.make_params(19, K, 768))
    x = torch.empty(M, K, device=DEV)
    for lo in range(0, M, 200000):
        x[lo:lo + 200000].normal_()
    f16 = ops.FoldedInProj([m.visual_input_proj, m.exp_visual_input_proj])
    f3 = ops.FoldedInProjX3([m.visual_input_proj, m.exp_visual_input_proj])
    with torch.no_grad():
        for fn, fold, exact in ((ops.in_proj_h16, f16, False), (ops.in_proj_x3, f3, True)):
            ys = fn(x, fold)
            assert all(torch.isfinite(y).all().item() for y in ys)
            for lo in (0, 699_000, 1_398_000, M - 2_000):            # 2^31 elements = row 699,051; 2^32 = row 1,398,101
                sub = fn(x[lo:lo + 2_000].contiguous(), fold)
                for a, b in zip(ys, sub):
                    if exact:
                        assert torch.equal(a[lo:lo + 2_000], b)
                    else:
                        assert (a[lo:lo + 2_000] - b).abs().max().item() < 5e-5
            del ys
    del x
    torch.cuda.empty_cache()


def test_parity_encode_uses_the_fused_projection_and_matches_the_unfused_one():
    """encode_context / encode_query in parity mode, inference: fused_parity_input_proj on (default) vs off."""
    m = _model(3072, 768, synth.make_params(17, 3072, 768))
    g = torch.Generator().manual_seed(3)
    v = torch.nn.functional.normalize(torch.randn(5, 40, 3072, generator=g).abs(), dim=-1).to(DEV)
    vm = torch.ones(5, 40, device=DEV)
    q = torch.nn.functional.normalize(torch.randn(7, 12, 768, generator=g), dim=-1).to(DEV)
    qm = torch.ones(7, 12, device=DEV)
    with torch.no_grad():
        assert m._use_fast(v) and m._use_fast(q)
        a = m.encode_context(v, vm) + m.encode_query(q, qm)
        m.fused_parity_input_proj = False
        assert not m._use_fast(v)
        from dldkd_amd import ops
        ops.ROWS_X3 = False
        try:
            b = m.encode_context(v, vm) + m.encode_query(q, qm)
        finally:
            ops.ROWS_X3 = True
    for x, y in zip(a, b):
        assert (x - y).abs().max().item() <= 2e-5 * max(1.0, y.abs().max().item())
    m.train()
    assert not m._use_fast(v)


def test_fast_path_keeps_rank_parity():
    """End to end (towers + scorer) with fast_input_proj on: R@1/5/10/100 vs the fp32 oracle within the gate."""
    from dldkd_amd import eval as ev
    import types
    m = _model(3072, 768, synth.make_params(51, 3072, 768))
    vids, txts = synth.make_eval_sets(5, nv=64, caps=3, dv=3072, dq=768)
    opt = types.SimpleNamespace(eval_context_bsz=25, eval_query_bsz=50, num_workers=0, pin_memory=False, device=torch.device(DEV),
                                double_branch=True)
    with torch.no_grad():
        ctx0 = ev.compute_context_info(m, synth.ListDataset(list(vids)), opt)
        f0, a0, b0, metas = ev.score_queries(m, synth.ListDataset(list(txts)), opt, ctx0)
        m.fast_input_proj = True
        ctx1 = ev.compute_context_info(m, synth.ListDataset(list(vids)), opt)
        f1, a1, b1, _ = ev.score_queries(m, synth.ListDataset(list(txts)), opt, ctx1)
    assert (f0 - f1).abs().max().item() < 4e-3           # cosine scores, fp16-operand input projection vs fp32 (bf16 scorer in both)
    _, t2v = ev.get_gt(ctx0["video_metas"], metas)
    r0, r1 = ev.eval_q2m(-f0, t2v), ev.eval_q2m(-f1, t2v)
    # random-init weights: near-chance, near-tied rankings; allow two of 192 queries to cross a cut
    for x, y in zip(r0[:4], r1[:4]):
        assert abs(x - y) <= 1.05, (r0, r1)


@pytest.mark.parametrize("K,L", [(3072, 128), (1024, 64), (768, 32)])
def test_k4_row_groups_project_only_the_listed_groups(K, L):
    """dldkd_in_proj_h16_rows128_groups: the rows of the listed 32-row groups equal the dense projection (the k order is
    rotated per workgroup, so equality is to fp32 summation order), every other output row stays untouched."""
    from dldkd_amd import ops
    torch.manual_seed(3)
    layers = [torch.nn.Module() for _ in range(2)]
    for l in layers:
        l.LayerNorm = torch.nn.LayerNorm(K).to(DEV)
        l.net = torch.nn.Sequential(torch.nn.Dropout(0.0), torch.nn.Linear(K, 384).to(DEV))
        l.LayerNorm.weight.data.uniform_(0.5, 1.5); l.LayerNorm.bias.data.normal_(0, 0.1)
    fold = ops.FoldedInProj(layers)
    n = 37
    g = torch.Generator().manual_seed(K)
    lens = torch.randint(1, L + 1, (n,), generator=g)
    lens[0], lens[1] = L, 1
    x = torch.randn(n, L, K, generator=g).to(DEV)
    dense = ops.in_proj_h16(x, fold)
    groups_np = ops.plan_row_groups(lens.numpy(), L)
    assert len(groups_np) % 4 == 0 and len(groups_np) < n * (L // 32) + 4
    groups = torch.from_numpy(groups_np).to(DEV)
    L_ = __import__("dldkd_amd.native", fromlist=["x"]).lib()
    ys = [torch.full((n * L, 384), -7.0, device=DEV) for _ in range(2)]
    from dldkd_amd import native
    f = fold.get()
    native.check(L_.dldkd_in_proj_h16_rows128_groups(native.ptr(x.view(-1, K)), native.ptr(f.Wf), native.ptr(f.cs), native.ptr(f.bb),
                                                      native.ptr(ys[0]), native.ptr(ys[1]), n * L, K, 1e-5, 1, native.ptr(groups),
                                                      groups.numel(), native.stream()), "groups")
    torch.cuda.synchronize()
    listed = torch.zeros(n * L, dtype=torch.bool)
    for r in groups_np.tolist():
        listed[r:r + 32] = True
    for b in range(2):
        got, want = ys[b].cpu(), dense[b].view(-1, 384).cpu()
        assert (got[listed] - want[listed]).abs().max().item() <= 2e-5 * want.abs().max().item() + 1e-6
        assert (got[~listed] == -7.0).all()
    # through the Python wrapper too
    via = ops.in_proj_h16(x, fold, groups=groups)
    assert torch.equal(via[0].view(-1, 384).cpu()[listed], ys[0].cpu()[listed])


@pytest.mark.parametrize("K,shape", [(3072, (37, 128)), (1024, (9, 64)), (256, (130, 32)), (3072, (300, 96))])
def test_k4b_resident_rows_match_k4_and_fp64(K, shape):
    """K4b (in_proj_rows128b: bf16 rows + precomputed LayerNorm statistics, ops.ResidentRows) against K4 on the fp32 rows it
    was filled from, and against fp64 with the kernels' roundings (bf16 x, bf16 W' = gamma (.) W): the ragged table holds
    exactly the valid rows (bf16 RNE of the features, statistics within fp32 rounding of fp64), the two kernels agree to fp32
    summation order (the k-steps are visited in a rotated order per workgroup), a row range in the middle of the table is
    served from its own tile grid."""
    from dldkd_amd import ops, model_components as mc
    torch.manual_seed(K + shape[0])
    n, L = shape
    layers = [mc.LinearLayer(K, 384, layer_norm=True, dropout=0.0, relu=True).to(DEV) for _ in range(2)]
    for l in layers:
        torch.nn.init.normal_(l.LayerNorm.weight, 1.0, 0.2)
        torch.nn.init.normal_(l.LayerNorm.bias, 0.0, 0.2)
    fold = ops.FoldedInProj(layers)
    lens = torch.randint(0, L + 1, (n,)).numpy()
    lens[0] = L
    x = torch.randn(n, L, K, device=DEV) * (1 + torch.rand(n, L, 1, device=DEV)) + 0.3
    tab = ops.ResidentRows(K, DEV)
    tab.append(x[: n // 2], lens[: n // 2])                  # two batches: the second is appended behind the first
    tab.append(x[n // 2:], lens[n // 2:])
    rows = torch.cat([x[i, :lens[i]] for i in range(n)], 0).contiguous()
    assert tab.rows == rows.shape[0] and tab.lens == [int(v) for v in lens]
    assert tab.xb.dtype == torch.float16 and torch.equal(tab.xb[:tab.rows], rows.to(torch.float16))
    mu, var = rows.double().mean(1), rows.double().var(1, unbiased=False)
    assert (tab.mean[:tab.rows].double() - mu).abs().max() < 1e-6
    assert (tab.rstd[:tab.rows].double() * (var + 1e-5).sqrt() - 1).abs().max() < 1e-5
    with torch.no_grad():
        y = ops.in_proj_resident(tab, 0, tab.rows, fold)
        y_k4 = ops.in_proj_h16(rows, fold)
        lo, hi = 7, tab.rows - 3
        y_mid = ops.in_proj_resident(tab, lo, hi, fold)
    for b, l in enumerate(layers):
        Wp = (l.net[1].weight * l.LayerNorm.weight).to(torch.float16).double()
        ref = (rows.to(torch.float16).double() @ Wp.T - mu[:, None] * Wp.sum(1)[None]) / (var + 1e-5).sqrt()[:, None] \
            + (l.net[1].weight.double() @ l.LayerNorm.bias.double() + l.net[1].bias.double())
        ref = ref.clamp_min(0)
        scale = ref.abs().max().item()
        assert (y[b].double() - ref).abs().max().item() < 2e-5 * max(scale, 1.0)
        assert (y[b] - y_k4[b]).abs().max().item() < 1e-5 * max(scale, 1.0)
        assert (y_mid[b] - y[b][lo:hi]).abs().max().item() < 1e-5 * max(scale, 1.0)


def test_k4b_rejects_what_it_cannot_serve():
    from dldkd_amd import native, ops
    assert not ops.in_proj_rows_ok(128) and not ops.in_proj_rows_ok(3072 + 32) and ops.in_proj_rows_ok(256)
    tab = ops.ResidentRows(256, DEV)
    with pytest.raises(native.NativeError):
        tab.append(torch.zeros(2, 4, 128, device=DEV), [4, 4])           # wrong feature width
    with pytest.raises(native.NativeError):
        tab.append(torch.zeros(2, 4, 256, device=DEV), [4, 5])           # a length past the padded batch
    tab.append(torch.zeros(2, 4, 256, device=DEV), [4, 0])
    assert tab.rows == 4
