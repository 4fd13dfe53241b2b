"""GPU parity of the training path: loss kernels vs the reference's golden values (G3) and the oracle's
autograd gradients; encoder backward kernels vs torch autograd of the oracle; the whole
DLDKD.forward + backward vs the reference's losses and 74 gradients (G4).

Tolerances: losses 1e-4 relative (BASELINE.json north_star); gradients are compared with the oracle in
fp64 at 2e-3 of each tensor's scale (fp32 kernels; the reference's own fp32 gradients differ from fp64 by
1.3e-4, make_golden.py)."""
import numpy as np
import pytest
import torch

import dldkd_oracle as orc
import synth
from test_encoder_gpu import _model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _g3(golden_dir):
    g = np.load(f"{golden_dir}/g3_losses.npz")
    counts = list(g["counts"])
    labels = [i for i, c in enumerate(counts) for _ in range(c)]
    L = g["predict"].shape[1]
    mask = torch.from_numpy((np.arange(L)[None] < g["lens"][:, None]).astype(np.float32))
    return g, labels, mask


def _lab(labels):
    return torch.tensor(labels, dtype=torch.int32, device=DEV)


def _close(a, b, tol=1e-4):
    a, b = (float(v.reshape(-1)[0]) if isinstance(v, np.ndarray) else float(v) for v in (a, b))     # goldens are 0-d / 1-element arrays
    assert abs(a - b) <= tol * max(1.0, abs(b)), (a, b)


def _gclose(got, ref, tol=2e-3):
    got, ref = got.double().cpu(), ref.double().cpu()
    scale = ref.abs().max().clamp_min(1e-12)
    assert ((got - ref).abs().max() / scale).item() <= tol, ((got - ref).abs().max().item(), scale.item())


def test_kl_vs_golden_and_grad(golden_dir):
    from dldkd_amd import functional as F_
    g, labels, mask = _g3(golden_dir)
    p = torch.from_numpy(g["predict"]).permute(0, 2, 1).contiguous().to(DEV).requires_grad_(True)   # (Nq, Nv, L)
    t = torch.from_numpy(g["target"]).permute(0, 2, 1).contiguous().to(DEV)
    lens = torch.from_numpy(g["lens"]).int().to(DEV)
    kl = F_.kl_frame(p, t, _lab(labels), lens, 0.2)
    _close(kl, g["kl"])
    kl.backward()
    po = torch.from_numpy(g["predict"]).double().requires_grad_(True)
    orc.kl_frame_score(po, torch.from_numpy(g["target"]).double(), mask.double(), labels).backward()
    _gclose(p.grad.permute(0, 2, 1), po.grad)


@pytest.mark.parametrize("alpha", [0.0, 0.3, 0.8, 1.0])
@pytest.mark.parametrize("beta", [0.5, 0.8])
def test_nce_soft_vs_golden_and_grad(golden_dir, alpha, beta):
    from dldkd_amd import functional as F_
    g, labels, _ = _g3(golden_dir)
    lab = _lab(labels)
    raw, sims = torch.from_numpy(g["raw"]), torch.from_numpy(g["sims"])
    S = raw.to(DEV).requires_grad_(True)
    v = F_.nce_soft(lab, S, sims.to(DEV), alpha, beta)                 # teacher-style soft labels (no grad to T)
    _close(v, g[f"nce_soft_a{alpha}_b{beta}"])
    v.backward()
    So = raw.double().requires_grad_(True)
    orc.nce_soft(labels, So, sims.double(), alpha, beta).backward()
    _gclose(S.grad, So.grad)
    S2 = raw.to(DEV).requires_grad_(True)
    v2 = F_.nce_soft(lab, S2, S2, alpha, beta)                          # exploration style: T is S (grad through targets)
    _close(v2, g[f"nce_self_a{alpha}_b{beta}"])
    v2.backward()
    So2 = raw.double().requires_grad_(True)
    orc.nce_soft(labels, So2, So2, alpha, beta).backward()
    _gclose(S2.grad, So2.grad)


def test_nce_hard_vs_golden_and_grad(golden_dir):
    from dldkd_amd import functional as F_
    g, labels, _ = _g3(golden_dir)
    raw = torch.from_numpy(g["raw"])
    S = raw.to(DEV).requires_grad_(True)
    v = F_.nce_hard(_lab(labels), S)
    _close(v, g["nce_hard"])
    v.backward()
    So = raw.double().requires_grad_(True)
    orc.nce_hard(labels, So).backward()
    _gclose(S.grad, So.grad)


@pytest.mark.parametrize("hard", [0, 1])
def test_triplet_vs_golden_and_grad(golden_dir, hard):
    from dldkd_amd import functional as F_
    g, labels, _ = _g3(golden_dir)
    cos = torch.from_numpy(g["cos"])
    r_t2v = torch.from_numpy(g[f"trip_hard{hard}_r_t2v"])
    r_v2t = torch.from_numpy(g["trip_hard0_r_v2t"]) if not hard else None
    C = cos.to(DEV).requires_grad_(True)
    v = F_.triplet(C, _lab(labels), r_t2v.int().to(DEV), None if r_v2t is None else r_v2t.int().to(DEV), bool(hard), 0.1)
    _close(v, g[f"trip_hard{hard}"])
    v.backward()
    Co = cos.double().requires_grad_(True)
    orc.clip_triplet_loss(Co, labels, 0.1, bool(hard), r_v2t, r_t2v).backward()
    _gclose(C.grad, Co.grad)


def test_triplet_method_consumes_rng_like_reference(golden_dir):
    """DLDKD.get_clip_triplet_loss draws from torch's CPU generator exactly as model.py:366-380 does."""
    g, labels, _ = _g3(golden_dir)
    m = _model(1024, 1024, synth.make_params(2, 1024, 1024))
    for hard in (False, True):
        m.set_hard_negative(hard, 5)
        torch.manual_seed(77)
        v = m.get_clip_triplet_loss(torch.from_numpy(g["cos"]).to(DEV), labels)
        _close(v, g[f"trip_hard{int(hard)}"])


def test_encoder_backward_ops_vs_autograd():
    from dldkd_amd import functional as F_
    gen = torch.Generator().manual_seed(11)
    N, L, Din = 3, 13, 64
    x = torch.randn(N, L, Din, generator=gen)
    lens = torch.tensor([13, 4, 9])
    mask = (torch.arange(L).unsqueeze(0) < lens.unsqueeze(1)).float()
    p = {}
    def mk(name, *shape, s=0.1):
        p[name] = (torch.randn(*shape, generator=gen) * s)
    mk("proj.LayerNorm.weight", Din, s=0.3); p["proj.LayerNorm.weight"] += 1
    mk("proj.LayerNorm.bias", Din); mk("proj.net.1.weight", 384, Din); mk("proj.net.1.bias", 384)
    mk("pos.position_embeddings.weight", 20, 384); mk("pos.LayerNorm.weight", 384, s=0.3); p["pos.LayerNorm.weight"] += 1
    mk("pos.LayerNorm.bias", 384)
    for n_ in ("query", "key", "value"):
        mk(f"enc.self.{n_}.weight", 384, 384, s=0.05); mk(f"enc.self.{n_}.bias", 384)
    mk("enc.output.dense.weight", 384, 384, s=0.05); mk("enc.output.dense.bias", 384)
    mk("enc.output.LayerNorm.weight", 384, s=0.3); p["enc.output.LayerNorm.weight"] += 1; mk("enc.output.LayerNorm.bias", 384)
    mk("w", 1, 384)
    # oracle (fp64 autograd)
    po = {k: v.double().requires_grad_(True) for k, v in p.items()}
    xo = x.double().requires_grad_(True)
    ho = orc.encode_input(xo, mask.double(), po, "proj", "enc", "pos", 4)
    out_o = orc.modular_pool(ho, mask.double(), po["w"])
    cot = torch.randn(N, 384, generator=gen)
    (out_o * cot.double()).sum().backward()
    # HIP
    pg = {k: v.to(DEV).requires_grad_(True) for k, v in p.items()}
    xg = x.to(DEV).requires_grad_(True)
    mg = mask.to(DEV)
    h = F_.layernorm(xg, pg["proj.LayerNorm.weight"], pg["proj.LayerNorm.bias"])
    h = F_.linear(h, pg["proj.net.1.weight"], pg["proj.net.1.bias"], relu=True)
    h = F_.layernorm(h, pg["pos.LayerNorm.weight"], pg["pos.LayerNorm.bias"], add=pg["pos.position_embeddings.weight"][:L], add_mod=L)
    w = torch.cat([pg[f"enc.self.{n_}.weight"] for n_ in ("query", "key", "value")], 0)
    b = torch.cat([pg[f"enc.self.{n_}.bias"] for n_ in ("query", "key", "value")], 0)
    ctxl = F_.attention(F_.linear(h, w, b), mg)
    h2 = F_.layernorm(F_.linear(ctxl, pg["enc.output.dense.weight"], pg["enc.output.dense.bias"]),
                      pg["enc.output.LayerNorm.weight"], pg["enc.output.LayerNorm.bias"], add=h, add_mod=0)
    out = F_.modpool(h2, mg, pg["w"].reshape(-1))
    assert (out.double().cpu() - out_o.detach()).abs().max() < 5e-5
    (out * cot.to(DEV)).sum().backward()
    _gclose(xg.grad, xo.grad)
    gmax = max(v.grad.abs().max().item() for v in po.values())
    for k in p:
        ref = po[k].grad
        got = pg[k].grad.double().cpu()
        assert (got - ref).abs().max().item() <= 2e-3 * max(ref.abs().max().item(), 1e-4 * gmax), k   # key bias: exactly-zero gradient, fp32 noise


def test_get_sim_scores_api_shapes_and_values(golden_dir):
    """The drop-in static methods: (pooled, clip_level (Nq, L, Nv)) with -1e10 on padded clips."""
    from dldkd_amd.model import DLDKD
    g = np.load(f"{golden_dir}/g1_simpool.npz")
    rs = np.random.RandomState(11)
    q = torch.from_numpy(rs.standard_normal((7, 384)).astype(np.float32))
    ctx = torch.from_numpy(rs.standard_normal((5, 9, 384)).astype(np.float32))
    mask = torch.from_numpy((np.arange(9)[None] < g["lens"][:, None]).astype(np.float32))
    ctx = ctx * mask.unsqueeze(-1)
    pooled, clip = DLDKD.get_sim_scores(q.to(DEV), ctx.to(DEV), mask.to(DEV))
    assert clip.shape == (7, 9, 5)
    assert np.abs(pooled.cpu().numpy() - g["pooled"]).max() < 2e-6
    assert np.abs(clip.cpu().numpy() - g["clip"]).max() < 2e-6 * 1e10 and (clip.cpu().numpy()[:, 3:, 1] == -1e10).all()
    valid = g["clip"] > -1e9
    assert np.abs(clip.cpu().numpy()[valid] - g["clip"][valid]).max() < 2e-6
    raw = DLDKD.get_unnormalized_sim_scores(q.to(DEV), ctx.to(DEV), mask.to(DEV))
    assert np.abs(raw.cpu().numpy() - g["raw"]).max() < 2e-5 * np.abs(g["raw"]).max()
    pooled_nm, _ = DLDKD.get_sim_scores(q.to(DEV), ctx.to(DEV))
    assert np.abs(pooled_nm.cpu().numpy() - g["pooled_nomask"]).max() < 2e-6


@pytest.mark.parametrize("tag,label_style,hard,caps", [("soft_rand", "soft", False, 1), ("soft_hard", "soft", True, 3),
                                                       ("hard_hard", "hard", True, 1)])
def test_forward_backward_vs_golden_g4(golden_dir, tag, label_style, hard, caps):
    g = np.load(f"{golden_dir}/g4_forward.npz")
    m = _model(3072, 768, synth.make_params(41, 3072, 768))
    m.label_style = label_style
    m.set_hard_negative(hard, 20)
    m.weight = 0.95 ** 2
    batch = synth.make_train_batch(1, nv=64, caps=caps, L=16, dv=3072, dq=768)
    batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    torch.manual_seed(4242)                               # same CPU RNG state the reference had
    loss, d = m(batch)
    for k in ("inher_trip", "inher_nce", "explore_trip", "explore_nce", "kl_intra"):
        _close(d[k], g[f"{tag}_{k}"])                     # 1e-4 relative
    _close(loss, g[f"{tag}_loss"])
    assert isinstance(d["loss_overall"], float)
    m.zero_grad()
    loss.backward()
    names = [n for n, _ in m.named_parameters()]
    assert len(names) == 74
    # A pre-activation within fp32 rounding of zero can land on the other side of the ReLU than in the
    # reference (measured: 1 of 393,216 at |pre| = 4.9e-7 in one configuration).  That is a discontinuity of
    # the function, not an error of the kernels: find such flips against the fp64 oracle and relax ONLY the
    # gradients of that tower's input projection (the flip changes one row's contribution to its
    # weight / bias / LayerNorm gradients by ~1e-2).
    p64 = {k: v.double() for k, v in synth.make_params(41, 3072, 768).items()}
    flipped = set()
    for pre, key in (("", "student_videos"), ("exp_", "student_videos"), ("", "student_text"), ("exp_", "student_text")):
        tower = pre + ("visual" if key == "student_videos" else "query") + "_input_proj"
        x = batch[key].cpu().double()
        ref_pre = orc._ln(x, p64[tower + ".LayerNorm.weight"], p64[tower + ".LayerNorm.bias"]) @ p64[tower + ".net.1.weight"].t() \
            + p64[tower + ".net.1.bias"]
        with torch.no_grad():
            ours = getattr(m, tower)(batch[key]).cpu()
        if bool(((ours > 0) != (ref_pre > 0)).any()):
            flipped.add(tower)
    gmax = max(float(np.abs(g[f"{tag}_grad/{n}/sample"]).max()) for n in names)
    nmax = max(float(g[f"{tag}_grad/{n}/norm"]) for n in names)
    for n, prm in m.named_parameters():
        tol = 5e-2 if any(n.startswith(t + ".") for t in flipped) else 3e-3
        gr = prm.grad.detach().reshape(-1).cpu()
        idx = np.unique(np.linspace(0, gr.numel() - 1, min(48, gr.numel())).astype(np.int64))
        ref = g[f"{tag}_grad/{n}/sample"].astype(np.float64)
        scale = max(np.abs(ref).max(), 1e-4 * gmax)       # key biases: exactly-zero gradient, fp32 noise
        assert np.abs(gr[idx].double().numpy() - ref).max() <= tol * scale, n
        rn = float(g[f"{tag}_grad/{n}/norm"])
        assert abs(float(gr.double().norm()) - rn) <= tol * max(rn, 1e-4 * nmax), n


@pytest.mark.parametrize("n,p", [(1, 0.5), (1027, 0.1), (4 * 128 * 128 * 16, 0.2)])
def test_fused_dropout_mask_scale_and_rng(n, p):
    """nn.Dropout semantics from one kernel: keep-rate 1-p, kept values scaled by 1/(1-p), the backward pass
    re-uses the same mask, and torch's CUDA generator governs the stream (manual_seed reproduces, calls differ)."""
    from dldkd_amd import functional as F_
    x = torch.randn(n, device=DEV).abs() + 0.5
    torch.manual_seed(11)
    a = x.clone().requires_grad_(True)
    y1 = F_.dropout(a, p, True)
    y2 = F_.dropout(x, p, True)                      # next slot of the generator: different mask
    torch.manual_seed(11)
    y3 = F_.dropout(x, p, True)
    assert torch.equal(y1.detach(), y3)
    kept = y1.detach() != 0
    assert torch.allclose(y1.detach()[kept], (x / (1 - p))[kept], rtol=1e-6)
    if n > 1000:
        assert not torch.equal(y1.detach(), y2)
        rate = kept.float().mean().item()
        assert abs(rate - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5 + 1e-3, rate
        # no short-period structure: keep flags of neighbouring elements are uncorrelated
        k = kept.float() - (1 - p)
        assert abs((k[1:] * k[:-1]).mean().item()) < 5 * p * (1 - p) / n ** 0.5 + 1e-3
    y1.backward(torch.ones_like(y1))
    assert torch.equal(a.grad != 0, kept)
    assert torch.allclose(a.grad[kept], torch.full_like(a.grad[kept], 1 / (1 - p)))
    assert F_.dropout(x, p, False) is x and F_.dropout(x, 0.0, True) is x


def test_forward_at_c3_size_vs_oracle():
    """BASELINE configs[2] size (128 videos x 5 captions = 640 queries, <=128 clips, 3072/768-d features, soft
    labels, hard negatives): the 7 losses of DLDKD.forward against the fp32 oracle on the same tensors and the same
    CPU random draws, 1e-4 relative (north_star).  Dropout off (model.eval()), like golden G4."""
    m = _model(3072, 768, synth.make_params(43, 3072, 768))
    m.label_style = "soft"
    m.set_hard_negative(True, 20)
    m.weight = 0.1 * 0.95 ** 0 / 0.1                      # train.py:76-80 at epoch 0
    batch = synth.make_train_batch(3, nv=128, caps=5, L=128, len_lo=24, dv=3072, dq=768)
    labels = batch["text_labels"]
    torch.manual_seed(99)
    rnd = [orc.draw_triplet_randoms(labels, 128, True, 20) for _ in range(2)]
    p = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    cfg = dict(n_heads=4, margin=0.1, use_hard_negative=True, label_style="soft", kl_intra_weight=0.1, weight=m.weight,
               inher_nce_weight=0.04, explore_nce_weight=0.04, alpha=0.8, belta=0.8)
    with torch.no_grad():
        ref = orc.forward_losses(p, batch, cfg, rnd)
    dbatch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    torch.manual_seed(99)
    loss, d = m(dbatch)
    for k in ("inher_trip", "inher_nce", "explore_trip", "explore_nce", "kl_intra"):
        _close(d[k], ref[k])
    _close(loss, ref["loss"])
    loss.backward()
    assert all(prm.grad is not None and torch.isfinite(prm.grad).all() for prm in m.parameters())


def test_forward_at_c3_size_bf16_mode_vs_oracle():
    """BASELINE configs[2] says bf16: the SAME C3-size step with every GEMM on bf16 MFMA (ops.set_gemm_precision("bf16"), what the
    6.6-ms c3_train_step_ms_bf16 number runs) against the fp32 oracle at the mode's tolerance, 2e-2 relative per loss term
    (the 1e-4 gate of north_star is met by the parity mode above; round 2 checked the bf16 mode at C1 size only)."""
    from dldkd_amd import ops
    m = _model(3072, 768, synth.make_params(43, 3072, 768))
    m.label_style = "soft"
    m.set_hard_negative(True, 20)
    m.weight = 1.0
    batch = synth.make_train_batch(3, nv=128, caps=5, L=128, len_lo=24, dv=3072, dq=768)
    labels = batch["text_labels"]
    torch.manual_seed(99)
    rnd = [orc.draw_triplet_randoms(labels, 128, True, 20) for _ in range(2)]
    p = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    cfg = dict(n_heads=4, margin=0.1, use_hard_negative=True, label_style="soft", kl_intra_weight=0.1, weight=m.weight,
               inher_nce_weight=0.04, explore_nce_weight=0.04, alpha=0.8, belta=0.8)
    with torch.no_grad():
        ref = orc.forward_losses(p, batch, cfg, rnd)
    dbatch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    ops.set_gemm_precision("bf16")
    try:
        torch.manual_seed(99)
        loss, d = m(dbatch)
        loss.backward()
    finally:
        ops.set_gemm_precision("fp32")
    for k in ("inher_trip", "inher_nce", "explore_trip", "explore_nce", "kl_intra"):
        a, b = float(d[k]), float(ref[k])
        assert abs(a - b) <= 2e-2 * max(abs(b), 1e-3) + 2e-3, (k, a, b)
    assert abs(float(loss) - float(ref["loss"])) <= 2e-2 * abs(float(ref["loss"]))
    assert all(prm.grad is not None and torch.isfinite(prm.grad).all() for prm in m.parameters())


@pytest.mark.parametrize("hard", [True, False])
def test_forward_at_c5_size_vs_oracle(hard):
    """BASELINE configs[4], one rank's step (Charades-STA: 128 videos, captions per video [3, 2, 2, ...] = 257 queries,
    <= 64 clips, 1024-d student features for both modalities, /root/reference/do_charades.sh:6-14): the 7 losses of
    DLDKD.forward against the fp32 oracle on the same tensors and the same CPU random draws, 1e-4 relative (north_star),
    plus finite gradients on all 74 parameters.  Dropout off, like golden G4 (its Philox masks have no reference twin)."""
    m = _model(1024, 1024, synth.make_params(45, 1024, 1024))
    m.label_style = "soft"
    m.set_hard_negative(hard, 20)
    m.weight = 1.0
    caps = sorted([3] + [2] * 127, reverse=True)
    batch = synth.make_train_batch(5, nv=128, caps=caps, L=64, len_lo=8, dv=1024, dq=1024)
    labels = batch["text_labels"]
    assert len(labels) == 257
    torch.manual_seed(55)
    rnd = [orc.draw_triplet_randoms(labels, 128, hard, 20) for _ in range(2)]
    p = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    cfg = dict(n_heads=4, margin=0.1, use_hard_negative=hard, label_style="soft", kl_intra_weight=0.1, weight=m.weight,
               inher_nce_weight=0.04, explore_nce_weight=0.04, alpha=0.8, belta=0.8)
    with torch.no_grad():
        ref = orc.forward_losses(p, batch, cfg, rnd)
    dbatch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    torch.manual_seed(55)
    loss, d = m(dbatch)
    for k in ("inher_trip", "inher_nce", "explore_trip", "explore_nce", "kl_intra"):
        _close(d[k], ref[k])
    _close(d["kl"], ref["kl_intra"])
    _close(loss, ref["loss"])
    loss.backward()
    assert all(prm.grad is not None and torch.isfinite(prm.grad).all() for prm in m.parameters())


@pytest.mark.parametrize("prec,tol,gtol", [("fp32", 2e-6, 2e-5), ("bf16", 2e-2, 3e-2)])
@pytest.mark.parametrize("nq,nv,L,D", [(640, 128, 128, 384), (257, 128, 64, 384), (37, 9, 50, 512), (5, 3, 1, 384), (130, 7, 128, 384)])
def test_fused_training_simpool_vs_fp64(prec, tol, gtol, nq, nv, L, D):
    """simpool_train.hip (one pooled MFMA GEMM forward, two gather kernels backward) against the reference formulas of
    get_sim_scores / get_unnormalized_sim_scores / the [i, :, label_i] read of compute_kl_loss (model.py:307-350, 184) in
    fp64 autograd: pooled cosine + raw maxima, the positive clip column, and the gradients of a random linear functional
    of all three with respect to queries and gallery.  Includes a video without valid clips and ragged lengths."""
    from dldkd_amd import functional as F_
    from dldkd_amd import ops
    g_ = torch.Generator().manual_seed(nq * 7 + nv)
    q = torch.randn(nq, D, generator=g_)
    g = torch.randn(nv, L, D, generator=g_)
    lens = torch.randint(1, L + 1, (nv,), generator=g_)
    if nv > 4:
        lens[3] = 0                                      # a video without clips: pooled -1e10, no gradient
    labels = torch.randint(0, nv, (nq,), generator=g_)
    labels[labels == 3] = 0
    mask = (torch.arange(L)[None] < lens[:, None])
    g = g * mask[..., None]
    Wc, Wr, We = torch.randn(nq, nv, generator=g_), torch.randn(nq, nv, generator=g_), torch.randn(nq, L, generator=g_)

    def reference(q64, g64):
        S = torch.einsum("nd,vld->nvl", q64, g64)
        qn = torch.nn.functional.normalize(q64, dim=-1)
        gn = torch.nn.functional.normalize(g64, dim=-1)
        C = torch.einsum("nd,vld->nvl", qn, gn)
        neg = torch.full_like(S, -1e10)
        Sm, Cm = torch.where(mask[None], S, neg), torch.where(mask[None], C, neg)
        pr, pc = Sm.max(-1).values, Cm.max(-1).values
        clip = Cm[torch.arange(nq), labels]                               # (nq, L)
        return pc, pr, clip
    q64, g64 = q.double().requires_grad_(), g.double().requires_grad_()
    pc64, pr64, clip64 = reference(q64, g64)
    valid_pos = mask[labels]
    (pc64 * Wc.double()).sum().add((pr64 * Wr.double()).sum()).add((clip64 * We.double() * valid_pos).sum()).backward()

    ops.set_gemm_precision(prec)
    try:
        qd, gd = q.to(DEV).requires_grad_(), g.to(DEV).requires_grad_()
        pc, pr, clip = F_.simpool_train(qd, gd, lens.to(DEV).int(), labels.to(DEV).int(), True)
        ((pc * Wc.to(DEV)).sum() + (pr * Wr.to(DEV)).sum() + (clip * (We * valid_pos).to(DEV)).sum()).backward()
    finally:
        ops.set_gemm_precision("fp32")
    has = lens > 0
    sc = max(1.0, float(pr64.detach()[:, has].abs().max()))
    assert (pr.detach().cpu().double()[:, has] - pr64.detach()[:, has]).abs().max().item() <= tol * sc
    assert (pc.detach().cpu().double()[:, has] - pc64.detach()[:, has]).abs().max().item() <= tol
    assert bool((pr.detach().cpu()[:, ~has] == -1e10).all()) and bool((pc.detach().cpu()[:, ~has] == -1e10).all())
    cm = clip.detach().cpu().double()
    assert (cm - clip64.detach())[valid_pos].abs().max().item() <= tol
    assert bool((cm[~valid_pos] == -1e10).all())
    if prec == "fp32":                # same arg-max clips as fp64 -> gradients agree to fp32 rounding
        _gclose(qd.grad, q64.grad, gtol)
        _gclose(gd.grad, g64.grad, gtol)
    else:                             # bf16 products can pick a different clip at near-ties: compare in aggregate
        for a, b in ((qd.grad, q64.grad), (gd.grad, g64.grad)):
            a, b = a.double().cpu().flatten(), b.flatten()
            assert torch.dot(a, b) / (a.norm() * b.norm()) > 0.98


@pytest.mark.parametrize("N,L,p_drop", [(6, 40, 0.0), (5, 128, 0.2), (9, 30, 0.15), (3, 1, 0.0), (4, 97, 0.3), (130, 32, 0.1)])
def test_fused_training_attention_vs_fp64(N, L, p_drop):
    """attention_train.hip (one forward kernel, two backward kernels, exact fp32 MFMA products) against the reference math of
    BertSelfAttention.forward (model_components.py:398-436) in fp64 autograd, with the dropout mask taken from the standalone
    dropout kernel on a tensor of the probabilities' shape at the same Philox (seed, offset) - the fused kernels must draw
    exactly those keep bits (also when L is not a multiple of 4 and a Philox call straddles two rows)."""
    from dldkd_amd import functional as F_
    g = torch.Generator().manual_seed(N * 131 + L)
    qkv = (torch.randn(N, L, 1152, generator=g) * 0.5)
    mask = torch.ones(N, L)
    if L > 3:
        mask[0, L // 2:] = 0
        mask[N - 1, L - 1:] = 0
    w = torch.randn(N, L, 384, generator=g)
    torch.manual_seed(1234)
    keep = torch.ones(N, 4, L, L)
    if p_drop > 0:
        _, kb = F_._dropout_fwd(torch.ones(N, 4, L, L, device=DEV), p_drop)      # the unfused kernel's mask for this slot
        keep = kb.float().cpu()
    q64 = qkv.double().requires_grad_()
    x = q64.view(N, L, 3, 4, 96)
    Q, K, V = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))                  # (N, 4, L, 96)
    S = Q @ K.transpose(-1, -2) / 96 ** 0.5 + ((1.0 - mask.double()) * -10000.0)[:, None, None, :]
    P = torch.softmax(S, -1)
    Pd = P * keep.double() / (1.0 - p_drop)
    ref = (Pd @ V).permute(0, 2, 1, 3).reshape(N, L, 384)
    (ref * w.double()).sum().backward()
    torch.manual_seed(1234)                                                        # same generator slot as the mask above
    a = qkv.to(DEV).requires_grad_()
    out = F_.attention(a, mask.to(DEV), p_drop, True)
    (out * w.to(DEV)).sum().backward()
    sc = ref.detach().abs().max().item()
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() <= 3e-6 * max(1.0, sc)
    _gclose(a.grad, q64.grad, 2e-5)


@pytest.mark.parametrize("N,L,p_drop", [(6, 40, 0.0), (5, 128, 0.2), (9, 30, 0.15), (3, 1, 0.0), (4, 97, 0.3), (130, 32, 0.1),
                                        (7, 64, 0.2), (3, 33, 0.0), (4, 80, 0.1)])
def test_bf16_training_attention_vs_fp64(N, L, p_drop):
    """attention_train_bf16.hip (throughput mode: one forward and ONE backward kernel, every product on the bf16 matrix cores,
    probabilities recomputed in the backward pass, four (sequence, head) pairs per workgroup when L <= 32) against the same fp64
    reference and the same dropout mask as the exact kernels, at the bf16 mode's tolerance (operands rounded to 8 bits of
    mantissa: 2e-2 of the largest entry)."""
    from dldkd_amd import functional as F_
    from dldkd_amd import ops
    g = torch.Generator().manual_seed(N * 131 + L)
    qkv = (torch.randn(N, L, 1152, generator=g) * 0.5)
    mask = torch.ones(N, L)
    if L > 3:
        mask[0, L // 2:] = 0
        mask[N - 1, L - 1:] = 0
    w = torch.randn(N, L, 384, generator=g)
    torch.manual_seed(1234)
    keep = torch.ones(N, 4, L, L)
    if p_drop > 0:
        _, kb = F_._dropout_fwd(torch.ones(N, 4, L, L, device=DEV), p_drop)
        keep = kb.float().cpu()
    q64 = qkv.double().requires_grad_()
    x = q64.view(N, L, 3, 4, 96)
    Q, K, V = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    S = Q @ K.transpose(-1, -2) / 96 ** 0.5 + ((1.0 - mask.double()) * -10000.0)[:, None, None, :]
    P = torch.softmax(S, -1)
    Pd = P * keep.double() / (1.0 - p_drop)
    ref = (Pd @ V).permute(0, 2, 1, 3).reshape(N, L, 384)
    (ref * w.double()).sum().backward()
    ops.set_gemm_precision("bf16")
    try:
        torch.manual_seed(1234)
        a = qkv.to(DEV).requires_grad_()
        out = F_.attention(a, mask.to(DEV), p_drop, True)
        assert type(out.grad_fn).__name__ == "_AttentionTrainBf16Backward"
        (out * w.to(DEV)).sum().backward()
    finally:
        ops.set_gemm_precision("fp32")
    sc = ref.detach().abs().max().item()
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() <= 2e-2 * max(1.0, sc)
    gd, g64 = a.grad.cpu().double(), q64.grad
    assert torch.isfinite(gd).all()
    assert (gd - g64).abs().max().item() <= 2e-2 * g64.abs().max().item()
    for blk in range(3):                                   # dq, dk, dv each on its own scale
        x_, y_ = gd[..., 384 * blk:384 * (blk + 1)].flatten(), g64[..., 384 * blk:384 * (blk + 1)].flatten()
        if y_.norm() > 0:
            assert torch.dot(x_, y_) / (x_.norm() * y_.norm()) > 0.999


@pytest.mark.parametrize("L", [8, 30, 32, 64, 128])
def test_bf16_training_attention_draws_the_exact_dropout_mask(L):
    """The keep bits, read back exactly: q = k = 0 makes P uniform (1/L), p = 0.5 makes Pd = 2/L (a power of two for these L,
    or exact in bf16), V[key][0] = 2^(key % 8) in column (key // 8) decodes the forward mask from the context; dO[q][0] =
    2^(q % 8) in column (q // 8) decodes the backward pass's mask (phase 2 reads the bits phase 1 left in LDS) from dV."""
    from dldkd_amd import functional as F_
    from dldkd_amd import ops
    N, p_drop = 3, 0.5
    qkv = torch.zeros(N, L, 3, 4, 96)
    for key in range(L):
        qkv[:, key, 2, :, key // 8] = 2.0 ** (key % 8)
    torch.manual_seed(99)
    _, kb = F_._dropout_fwd(torch.ones(N, 4, L, L, device=DEV), p_drop)
    keep = kb.float().cpu()                                                     # (N, 4, q, key)
    w = torch.zeros(N, L, 4, 96)
    for q in range(L):
        w[:, q, :, q // 8] = 2.0 ** (q % 8)
    ops.set_gemm_precision("bf16")
    try:
        torch.manual_seed(99)
        a = qkv.reshape(N, L, 1152).to(DEV).requires_grad_()
        out = F_.attention(a, None, p_drop, True)
        (out * w.reshape(N, L, 384).to(DEV)).sum().backward()
    finally:
        ops.set_gemm_precision("fp32")
    pd = 2.0 / L
    exp_fwd = torch.zeros(N, L, 4, 96)
    for c in range((L + 7) // 8):
        ks = slice(8 * c, min(8 * c + 8, L))
        wt = 2.0 ** torch.arange(ks.stop - ks.start).float()
        exp_fwd[:, :, :, c] = (keep[:, :, :, ks] * wt).sum(-1).permute(0, 2, 1) * pd
    got = out.detach().cpu().view(N, L, 4, 96)
    tol = 0.0 if (L & (L - 1)) == 0 else 1e-2        # 2 / 30 is not a bf16 number
    assert (got - exp_fwd).abs().max().item() <= tol * exp_fwd.abs().max().item()
    # dV[key][head, c] = sum_q Pd[q][key] dO[q][head, c] = pd * sum_{q in 8c..8c+7} keep[q][key] 2^(q % 8)
    exp_dv = torch.zeros(N, L, 4, 96)
    for c in range((L + 7) // 8):
        qs = slice(8 * c, min(8 * c + 8, L))
        wt = 2.0 ** torch.arange(qs.stop - qs.start).float()
        exp_dv[:, :, :, c] = (keep[:, :, qs, :] * wt[None, None, :, None]).sum(2).permute(0, 2, 1) * pd
    dv = a.grad.cpu().view(N, L, 3, 4, 96)[:, :, 2]
    assert (dv - exp_dv).abs().max().item() <= tol * exp_dv.abs().max().item()


@pytest.mark.parametrize("M,D,with_pos", [(300, 3072, False), (257, 768, False), (40 * 30, 384, True), (5, 1024, False)])
def test_fused_layernorm_dropout_equals_layernorm_then_dropout(M, D, with_pos):
    """LayerNorm -> Dropout as ONE kernel (LinearLayer / TrainablePositionalEncoding in training, model_components.py:277-312)
    against the two separate kernels at the same Philox slot: identical outputs, and identical gradients for x, gamma, beta and
    the position table (the fused backward masks dy inside the LayerNorm backward)."""
    from dldkd_amd import functional as F_
    g = torch.Generator().manual_seed(M + D)
    x = torch.randn(M, D, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    pos = torch.randn(30, D, generator=g) if with_pos else None
    w = torch.randn(M, D, generator=g).to(DEV)
    res = []
    for fused in (True, False):
        xs, gs, bs = (t.to(DEV).requires_grad_() for t in (x, gamma, beta))
        ps = pos.to(DEV).requires_grad_() if with_pos else None
        xin = xs.view(M // 30, 30, D) if with_pos else xs
        torch.manual_seed(77)
        if fused:
            y = F_.layernorm(xin, gs, bs, add=ps, add_mod=30 if with_pos else 0, p_drop=0.2, training=True)
        else:
            y = F_.dropout(F_.layernorm(xin, gs, bs, add=ps, add_mod=30 if with_pos else 0), 0.2, True)
        (y.reshape(M, D) * w).sum().backward()
        res.append((y.detach().reshape(M, D), xs.grad, gs.grad, bs.grad, ps.grad if with_pos else None))
    assert torch.equal(res[0][0], res[1][0])
    frac = (res[0][0] == 0).float().mean().item()
    assert 0.15 < frac < 0.25
    assert torch.equal(res[0][1], res[1][1])                      # dx: same arithmetic on the same masked dy
    for a, b in zip(res[0][2:], res[1][2:]):                      # dgamma / dbeta / dpos accumulate with fp32 atomics
        if a is not None:
            _gclose(a, b, 1e-5)


def test_loss_kernels_fuzz_vs_oracle():
    """Random batch structures (2..40 videos, 1..4 captions each, 1..48 clips, random alpha/beta, both negative
    modes) through every loss kernel: value 1e-4 and gradient 2e-3 against the fp64 oracle."""
    from dldkd_amd import functional as F_
    rs = np.random.RandomState(77)
    for case in range(12):
        nv = int(rs.randint(2, 41))
        counts = sorted(rs.randint(1, 5, size=nv).tolist(), reverse=True)
        labels = [i for i, c in enumerate(counts) for _ in range(c)]
        nq, L = len(labels), int(rs.randint(1, 49))
        lens = rs.randint(1, L + 1, size=nv)
        mask = torch.from_numpy((np.arange(L)[None] < lens[:, None]).astype(np.float32))
        g = torch.Generator().manual_seed(1000 + case)
        lab = _lab(labels)
        # KL over the ground-truth video's clips
        p = torch.randn(nq, nv, L, generator=g) * 0.5
        t = torch.randn(nq, nv, L, generator=g) * 0.5
        pd = p.to(DEV).requires_grad_(True)
        kl = F_.kl_frame(pd, t.to(DEV), lab, torch.from_numpy(lens).int().to(DEV), 0.2)
        po = p.permute(0, 2, 1).double().requires_grad_(True)
        ko = orc.kl_frame_score(po, t.permute(0, 2, 1).double(), mask.double(), labels)
        _close(kl, ko)
        kl.backward()
        ko.backward()
        _gclose(pd.grad.permute(0, 2, 1), po.grad)
        # InfoNCE, soft and hard labels
        raw = torch.randn(nq, nv, generator=g) * 3.0
        sims = torch.randn(nq, nv, generator=g) * 3.0
        alpha, beta = float(rs.choice([0.0, 0.25, 0.8, 1.0])), float(rs.uniform(0.3, 0.9))
        S = raw.to(DEV).requires_grad_(True)
        v = F_.nce_soft(lab, S, sims.to(DEV), alpha, beta)
        So = raw.double().requires_grad_(True)
        vo = orc.nce_soft(labels, So, sims.double(), alpha, beta)
        _close(v, vo)
        v.backward()
        vo.backward()
        _gclose(S.grad, So.grad)
        S2 = raw.to(DEV).requires_grad_(True)
        v2 = F_.nce_hard(lab, S2)
        So2 = raw.double().requires_grad_(True)
        vo2 = orc.nce_hard(labels, So2)
        _close(v2, vo2)
        v2.backward()
        vo2.backward()
        _gclose(S2.grad, So2.grad)
        # triplet, both negative modes, with the reference's random draws
        cos = torch.tanh(torch.randn(nq, nv, generator=g))
        for hard in (False, True):
            torch.manual_seed(case)
            r_v2t, r_t2v = orc.draw_triplet_randoms(labels, nv, hard, 20)
            C = cos.to(DEV).requires_grad_(True)
            tv = F_.triplet(C, lab, r_t2v.int().to(DEV), None if r_v2t is None else r_v2t.int().to(DEV), hard, 0.1)
            Co = cos.double().requires_grad_(True)
            to = orc.clip_triplet_loss(Co, labels, 0.1, hard, r_v2t, r_t2v)
            _close(tv, to)
            tv.backward()
            to.backward()
            _gclose(C.grad, Co.grad)


@pytest.mark.parametrize("M,D", [(70, 64), (1000, 384), (257, 768), (300, 1024), (129, 1540), (200, 2048), (130, 3072), (65, 4096)])
def test_layernorm_forward_backward_all_widths(M, D):
    """Every template instantiation of the LayerNorm kernels (2 / 4 / 16 float4 per lane, 16- and 4-wave workgroups)
    against fp64 autograd, with and without the residual input."""
    from dldkd_amd import functional as F_
    g = torch.Generator().manual_seed(M + D)
    x, add = torch.randn(M, D, generator=g), torch.randn(M, D, generator=g)
    gam, bet = 1 + 0.3 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    cot = torch.randn(M, D, generator=g)
    for use_add in (False, True):
        xs = [t.to(DEV).requires_grad_(True) for t in (x, gam, bet, add)]
        y = F_.layernorm(xs[0], xs[1], xs[2], add=xs[3] if use_add else None, add_mod=0)
        (y * cot.to(DEV)).sum().backward()
        xo = [t.double().requires_grad_(True) for t in (x, gam, bet, add)]
        yo = torch.nn.functional.layer_norm(xo[0] + (xo[3] if use_add else 0), (D,), xo[1], xo[2], 1e-5)
        (yo * cot.double()).sum().backward()
        assert (y.double().cpu() - yo.detach()).abs().max().item() < 2e-5
        for a, b in zip(xs[:3] + ([xs[3]] if use_add else []), xo[:3] + ([xo[3]] if use_add else [])):
            _gclose(a.grad, b.grad, tol=1e-3)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_fused_training_simpool_nan_query_propagates_and_stays_in_bounds(prec):
    """an all-NaN query row (a diverged step) must come out as NaN scores - torch.max propagates NaN,
    model.py:327,349 - with arg-max indices INSIDE the video, so the backward gathers stay in bounds (the `s > best` pool left
    the 0x7fffffff sentinel there and the gather read 2^31 rows past the gallery: a GPU memory fault)."""
    from dldkd_amd import functional as F_
    from dldkd_amd import ops
    g_ = torch.Generator().manual_seed(5)
    nq, nv, L, D = 70, 9, 40, 384
    q = torch.randn(nq, D, generator=g_)
    g = torch.randn(nv, L, D, generator=g_)
    lens = torch.randint(1, L + 1, (nv,), generator=g_)
    g = g * (torch.arange(L)[None] < lens[:, None])[..., None]
    labels = torch.randint(0, nv, (nq,), generator=g_)
    q[7] = float("nan")                                      # every product of this query is NaN
    q[33, 5] = float("inf")
    ops.set_gemm_precision(prec)
    try:
        qd, gd = q.to(DEV).requires_grad_(), g.to(DEV).requires_grad_()
        pc, pr, clip = F_.simpool_train(qd, gd, lens.to(DEV).int(), labels.to(DEV).int(), True)
        (pc.sum() + pr.sum()).backward()
        torch.cuda.synchronize()                             # a wild gather would fault here
    finally:
        ops.set_gemm_precision("fp32")
    assert torch.isnan(pr[7]).all() and torch.isnan(pc[7]).all()          # NaN propagates like torch.max
    ok = torch.ones(nq, dtype=torch.bool)
    ok[[7, 33]] = False
    assert torch.isfinite(pr[ok.to(DEV)]).all() and torch.isfinite(pc[ok.to(DEV)]).all()
    assert torch.isfinite(qd.grad[ok.to(DEV)]).all()


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
@pytest.mark.parametrize("M,K,p_drop", [(300, 3072, 0.2), (257, 768, 0.15), (1000, 1024, 0.0), (5, 64, 0.3)])
def test_fused_training_input_projection_vs_unfused(M, K, p_drop, prec):
    """LinearLayer on raw features in training (throughput and parity mode) as ONE autograd node (functional._InProjTrain: the backward pass
    takes LayerNorm's parameter gradients from the accumulators of dy W and never forms the input gradient) against the unfused
    chain LayerNorm-dropout -> Linear (same kernels forward, dX GEMM + LayerNorm backward behind): identical forward, gradients
    of W / b equal, gradients of gamma / beta within bf16-product rounding of the unfused ones and of an fp64 evaluation of the
    same formula."""
    from dldkd_amd import functional as F_
    from dldkd_amd import ops
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g) * (1 + torch.rand(M, 1, generator=g))
    gamma, beta = 1 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    W, b = torch.randn(384, K, generator=g) * 0.02, 0.1 * torch.randn(384, generator=g)
    w_out = torch.randn(M, 384, generator=g).to(DEV)
    res = []
    ops.set_gemm_precision(prec)
    try:
        for fused in (True, False):
            F_.IN_PROJ_TRAIN_FUSED = fused
            gs, bs, Ws, bbs = (t.to(DEV).requires_grad_() for t in (gamma, beta, W, b))
            torch.manual_seed(55)
            xd = x.to(DEV)
            if fused:
                assert F_.in_proj_train_ok(xd, Ws)
                y = F_.in_proj_train(xd, gs, bs, Ws, bbs, p_drop, True)
                assert type(y.grad_fn).__name__ == "_InProjTrainBackward"
            else:
                y = F_.linear(F_.layernorm(xd, gs, bs, p_drop=p_drop, training=True), Ws, bbs, relu=True)
            (y * w_out).sum().backward()
            res.append((y.detach().cpu(), gs.grad.cpu(), bs.grad.cpu(), Ws.grad.cpu(), bbs.grad.cpu()))
    finally:
        F_.IN_PROJ_TRAIN_FUSED = True
        ops.set_gemm_precision("fp32")
    (yf, dgf, dbf, dWf, dbbf), (yu, dgu, dbu, dWu, dbbu) = res
    assert torch.equal(yf, yu)
    # (same kernels on the same operands; split-K planes and the column sums add with fp32 atomics, whose order varies)
    assert torch.allclose(dWf, dWu, rtol=1e-4, atol=1e-5 * dWu.abs().max().item())
    assert torch.allclose(dbbf, dbbu, rtol=1e-4, atol=1e-5 * dbbu.abs().max().item())
    # fp64 evaluation from the fused run's own forward quantities (mask recovered from the dropped activations)
    x64 = x.double()
    mu, var = x64.mean(1, keepdim=True), x64.var(1, unbiased=False, keepdim=True)
    xh = (x64 - mu) / torch.sqrt(var + 1e-5)
    dy = (w_out.cpu().double() * (yf > 0))
    dz = dy @ W.double()
    if p_drop > 0:
        torch.manual_seed(55)
        _, kb = F_._dropout_fwd(torch.ones(M, K, device=DEV), p_drop)
        dz = dz * kb.cpu().double() / (1 - p_drop)
    dg64, db64 = (dz * xh).sum(0), dz.sum(0)
    tol = 2e-2 if prec == "bf16" else 2e-5          # parity mode: fp32-grade products (three bf16 planes per operand)
    for got, ref in ((dgf, dg64), (dbf, db64), (dgu, dg64), (dbu, db64)):
        assert (got.double() - ref).abs().max().item() <= tol * ref.abs().max().item()


def test_video_tower_skips_the_padding_in_parity_mode():
    """The padding skip of the training video towers (input projection with the batch's mask -> 32-row group flags -> every row-wise
    kernel of the tower skips the groups without a valid clip) in parity mode: fp32 rows, three-plane GEMMs
    (dldkd_gemm_f32x3_flags: row tiles / k-tiles of the padding are not multiplied).  Losses and all 74 gradients equal those of
    the run that computes the padding, up to fp32 summation order, with dropout on."""
    import types
    import synth
    from dldkd_amd import functional as F_, ops
    from dldkd_amd.model import DLDKD
    cfg = types.SimpleNamespace(visual_input_size=512, query_input_size=256, inheritance_hidden=384, exploration_hidden=384,
                                max_ctx_l=64, max_desc_l=30, input_drop=0.2, drop=0.2, n_heads=4, initializer_range=0.02,
                                margin=0.1, use_hard_negative=False, hard_pool_size=20, label_style="soft")
    mopt = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                 collection="tvr", alpha=0.8, belta=0.8)
    batch = synth.make_train_batch(33, nv=48, caps=3, L=64, len_lo=5, dv=512, dq=256)
    batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    assert ops.gemm_precision() in ("fp32", "fp32x3") and batch["student_videos"].shape[1] == 64
    res, seen = [], []
    real = ops.row_groups
    try:
        ops.row_groups = lambda M: seen.append(real(M) is not None) or real(M)
        for skip in (True, False):
            F_.IN_PROJ_SKIP_PADDING = DLDKD.TOWER_SKIPS_PADDING = skip
            n0 = sum(seen)
            torch.manual_seed(5)
            m = DLDKD(types.SimpleNamespace(**vars(cfg)), mopt).to(DEV).train()
            torch.manual_seed(9)
            loss, parts = m(batch)
            loss.backward()
            assert (sum(seen) > n0) == skip                       # the tower's kernels did / did not get the flags
            res.append((float(loss), {n: p.grad.detach().clone() for n, p in m.named_parameters()}))
    finally:
        ops.row_groups = real
        F_.IN_PROJ_SKIP_PADDING = DLDKD.TOWER_SKIPS_PADDING = True
    (la, ga), (lb, gb) = res
    assert la == pytest.approx(lb, rel=2e-6)
    assert len(ga) == 74
    for n in ga:
        assert torch.isfinite(ga[n]).all(), n
        scale = gb[n].abs().max().item()
        assert (ga[n] - gb[n]).abs().max().item() <= 2e-5 * max(scale, 1e-6) + 1e-8, (n, scale)


@pytest.mark.parametrize("nq,Lq", [(640, 30), (257, 24), (1, 30), (1, 1), (5, 1), (7, 30), (642, 8), (3, 17)])
def test_modular_pooling_backward_vs_oracle_fp64(nq, Lq):
    """get_modularized_queries (method/model.py:245-258) backward: `modpool_bwd_kernel` (four queries per workgroup, the dw parts
    added in LDS before the atomics) against the fp64 oracle's autograd - at the TVR batch's 640 queries, at query counts that are
    not a multiple of 4, with one query, and at the word-count edges (1 word: the softmax is the constant 1 and dw = 0; 30 words =
    max_desc_l).  Run twice: dh must be bit-identical (no atomics in it), dw - accumulated with fp32 atomics across workgroups -
    equal up to the order of the sums."""
    from dldkd_amd import functional as F_
    gen = torch.Generator().manual_seed(1000 + 31 * nq + Lq)
    h = torch.randn(nq, Lq, 384, generator=gen)
    lens = torch.randint(1, Lq + 1, (nq,), generator=gen)
    lens[0] = Lq
    if nq > 2:
        lens[1] = 1
    mask = (torch.arange(Lq).unsqueeze(0) < lens.unsqueeze(1)).float()
    w = torch.randn(1, 384, generator=gen) * 0.2
    cot = torch.randn(nq, 384, generator=gen)
    ho, wo = h.double().requires_grad_(True), w.double().requires_grad_(True)
    out_o = orc.modular_pool(ho, mask.double(), wo)
    (out_o * cot.double()).sum().backward()
    res = []
    for _ in range(2):
        hg, wg = h.to(DEV).requires_grad_(True), w.reshape(-1).to(DEV).requires_grad_(True)
        out = F_.modpool(hg, mask.to(DEV), wg)
        assert (out.double().cpu() - out_o.detach().reshape(out.shape)).abs().max() < 2e-5
        (out * cot.to(DEV)).sum().backward()
        res.append((hg.grad.clone(), wg.grad.clone()))
    dh, dw = res[0]
    assert torch.equal(dh, res[1][0])
    href, wref = ho.grad, wo.grad.reshape(-1)
    assert (dh.double().cpu() - href).abs().max().item() <= 2e-5 * max(href.abs().max().item(), 1.0)
    # gradients of masked words are exactly zero, as the reference's mask_logits (-1e10) leaves them
    assert float((dh.cpu() * (1 - mask)[..., None]).abs().max()) == 0.0
    scale = max(wref.abs().max().item(), 1e-6)
    for got in (dw, res[1][1]):
        assert (got.double().cpu() - wref).abs().max().item() <= 2e-4 * scale + 1e-6
    assert (dw - res[1][1]).abs().max().item() <= 1e-5 * scale + 1e-7


def test_fused_branch_losses_scale_by_upstream_gradients_once():
    """functional._BranchLoss hands out gradients computed for an upstream gradient of 1 and scales them in the backward pass: with
    NON-unit upstream gradients the result must be the weighted sum of the terms' own gradients, and a second backward pass through
    the same node (retain_graph=True) - which would scale the saved tensors twice - raises."""
    from dldkd_amd import functional as F_
    rs = np.random.RandomState(5)
    nv = 12
    counts = sorted(rs.randint(1, 4, size=nv).tolist(), reverse=True)
    labels = [i for i, c in enumerate(counts) for _ in range(c)]
    nq, L = len(labels), 20
    lens_np = rs.randint(1, L + 1, size=nv)
    g = torch.Generator().manual_seed(42)
    lab = _lab(labels)
    lens = torch.from_numpy(lens_np).int().to(DEV)
    cos = torch.tanh(torch.randn(nq, nv, generator=g)).to(DEV)
    raw = (torch.randn(nq, nv, generator=g) * 3.0).to(DEV)
    tch = (torch.randn(nq, nv, generator=g) * 3.0).to(DEV)
    clip_p = (torch.randn(nq, L, generator=g) * 0.5).to(DEV)
    clip_t = (torch.randn(nq, L, generator=g) * 0.5).to(DEV)
    torch.manual_seed(3)
    r_v2t, r_t2v = orc.draw_triplet_randoms(labels, nv, True, 20)
    r_t2v = r_t2v.int().to(DEV)
    r_v2t = None if r_v2t is None else r_v2t.int().to(DEV)

    def run(weights, retain=False):
        C, S, P = cos.clone().requires_grad_(True), raw.clone().requires_grad_(True), clip_p.clone().requires_grad_(True)
        terms = F_.branch_losses(C, S, tch, P, clip_t, lab, lens, r_t2v, r_v2t, True, 0.1, True, 0.8, 0.8, 0.04, 0.1, False)
        total = sum(w_ * t for w_, t in zip(weights, terms) if w_ is not None)
        total.backward(retain_graph=retain)
        return [x.grad.clone() if x.grad is not None else torch.zeros_like(x) for x in (C, S, P)], total, [float(t) for t in terms]

    wts = (2.0, -3.0, 0.5)
    got, total, vals = run(wts, retain=True)
    parts = [run(tuple(1.0 if i == j else None for i in range(3)))[0] for j in range(3)]
    for k in range(3):
        ref = sum(w_ * parts[j][k] for j, w_ in enumerate(wts))
        assert (got[k] - ref).abs().max().item() <= 1e-6 * max(ref.abs().max().item(), 1e-3), k
    assert all(np.isfinite(v) for v in vals) and vals[0] > 0
    with pytest.raises(RuntimeError, match="second backward"):
        total.backward()


@pytest.mark.parametrize("soft", [True, False])
def test_branch_losses_read_the_epoch_schedule_from_device_words(soft):
    """functional.ScheduleWords: alpha (-> hardQ / hardV and the coefficient vectors), belta and the KD weight of an epoch
    (method/train.py:66-113) as device state the branch-loss launches read by address, so that a captured step follows the schedule
    without a new capture.  For three schedule points - the values rewritten IN PLACE between the calls, the by-value arguments left
    at the first epoch's - losses and gradients equal the by-value launch bit for bit; both branches' forms (teacher targets + KL,
    folded targets without KL)."""
    from dldkd_amd import functional as F_
    rs = np.random.RandomState(9)
    nv = 16
    counts = sorted(rs.randint(1, 4, size=nv).tolist(), reverse=True)
    labels = [i for i, c in enumerate(counts) for _ in range(c)]
    nq, L = len(labels), 24
    g = torch.Generator().manual_seed(43)
    lab = _lab(labels)
    lens = torch.from_numpy(rs.randint(1, L + 1, size=nv)).int().to(DEV)
    cos = torch.tanh(torch.randn(nq, nv, generator=g)).to(DEV)
    raw = (torch.randn(nq, nv, generator=g) * 3.0).to(DEV)
    tch = (torch.randn(nq, nv, generator=g) * 3.0).to(DEV)
    clip_p = (torch.randn(nq, L, generator=g) * 0.5).to(DEV)
    clip_t = (torch.randn(nq, L, generator=g) * 0.5).to(DEV)
    torch.manual_seed(4)
    r_v2t, r_t2v = orc.draw_triplet_randoms(labels, nv, True, 20)
    r_t2v = r_t2v.int().to(DEV)
    r_v2t = None if r_v2t is None else r_v2t.int().to(DEV)
    kl_factor = 0.1

    def run(alpha, belta, weight, sw, first):
        out = []
        for inher in (True, False):
            C, S, P = cos.clone().requires_grad_(True), raw.clone().requires_grad_(True), clip_p.clone().requires_grad_(True)
            a, b, w = (first if sw is not None else (alpha, belta, weight))      # by-value arguments: stale under the words
            with F_.schedule_words(sw):
                if inher:
                    terms = F_.branch_losses(C, S, tch, P, clip_t, lab, lens, r_t2v, r_v2t, True, 0.1, soft, a, b, 0.04, kl_factor * w, False,
                                             kd_factor=kl_factor)
                else:
                    terms = F_.branch_losses(C, S, None, None, None, lab, lens, r_t2v, r_v2t, True, 0.1, soft, a, b, 0.04, 0.0, True,
                                             kd_factor=0.0)[:2]
            sum(terms).backward()
            out.append([t.detach().clone() for t in terms] + [x.grad.clone() for x in (C, S, P) if x.grad is not None])
        return out

    sw = F_.ScheduleWords(nq, nv, soft, DEV)
    for f in (kl_factor, 0.0):
        sw.words_for(f)
    points = [(0.8, 0.8, 1.0), (0.37, 0.61, 0.95 ** 7), (0.0, 0.5, 0.05)]
    for i, (a, b, w) in enumerate(points):
        assert sw.update(a, b, w) == True and sw.update(a, b, w) == False       # noqa: E712  (rewritten once per change)
        got, ref = run(a, b, w, sw, points[0]), run(a, b, w, None, None)
        for x, y in zip(got, ref):
            assert len(x) == len(y) and all(torch.equal(p_, q_) for p_, q_ in zip(x, y)), i
    assert sw.used == 2 * len(points)
    if soft:       # the schedule does move the losses (the test would pass vacuously otherwise)
        assert float(run(*points[0], None, None)[0][1]) != float(run(*points[1], None, None)[0][1])


@pytest.mark.parametrize("soft,hard_neg", [(True, True), (True, False), (False, True)])
def test_branch_losses_skip_padding_queries(soft, hard_neg):
    """dldkd_branch_losses_f32 with nq_valid < nq (a query axis padded to a bucket, so that batches of different caption counts -
    method/data_provider.py:34-72 - share a captured graph): with GARBAGE in the padding rows of every input the three terms and the
    gradients of the real rows equal the launch on the unpadded matrices (to the last bits of a sum), and the padding rows' gradients are exactly zero;
    by value and through the device words (functional.ScheduleWords), for both branches' forms."""
    from dldkd_amd import functional as F_
    rs = np.random.RandomState(19)
    nv = 16
    counts = sorted(rs.randint(1, 4, size=nv).tolist(), reverse=True)
    labels = [i for i, c in enumerate(counts) for _ in range(c)]
    nq, L = len(labels), 24
    nq_b = -(-nq // 32) * 32
    assert nq_b > nq
    g = torch.Generator().manual_seed(47)
    lens = torch.from_numpy(rs.randint(1, L + 1, size=nv)).int().to(DEV)

    def padded(x, fill):
        out = torch.full((nq_b,) + tuple(x.shape[1:]), fill, dtype=x.dtype, device=DEV)
        out[:nq] = x
        return out
    cos = torch.tanh(torch.randn(nq, nv, generator=g)).to(DEV)
    raw = (torch.randn(nq, nv, generator=g) * 3.0).to(DEV)
    tch = (torch.randn(nq, nv, generator=g) * 3.0).to(DEV)
    clip_p = (torch.randn(nq, L, generator=g) * 0.5).to(DEV)
    clip_t = (torch.randn(nq, L, generator=g) * 0.5).to(DEV)
    torch.manual_seed(6)
    r_v2t, r_t2v = orc.draw_triplet_randoms(labels, nv, hard_neg, 20)
    r_t2v = r_t2v.int().to(DEV)
    r_v2t = None if r_v2t is None else r_v2t.int().to(DEV)
    lab = _lab(labels)
    lab_b = torch.zeros(nq_b, dtype=torch.int32, device=DEV)
    lab_b[:nq] = lab
    r_t2v_b = torch.ones(nq_b, dtype=torch.int32, device=DEV)
    r_t2v_b[:nq] = r_t2v

    def run(pad, sw):
        out = []
        for inher in (True, False):
            ins = [cos, raw, tch, clip_p, clip_t]
            if pad:      # garbage (huge scores, NaN clip rows) where the padding queries sit
                ins = [padded(cos, 50.0), padded(raw, 1e4), padded(tch, -1e4), padded(clip_p, float("nan")), padded(clip_t, float("nan"))]
            C, S, T, P, Pt = ins
            C, S, P = C.clone().requires_grad_(True), S.clone().requires_grad_(True), P.clone().requires_grad_(True)
            kw = dict(nq_valid=nq) if pad else {}
            with F_.schedule_words(sw):
                if inher:
                    terms = F_.branch_losses(C, S, T, P, Pt, lab_b if pad else lab, lens, r_t2v_b if pad else r_t2v, r_v2t, hard_neg, 0.1, soft,
                                             0.8, 0.7, 0.04, 0.1 * 0.9, False, kd_factor=0.1, **kw)
                else:
                    terms = F_.branch_losses(C, S, None, None, None, lab_b if pad else lab, lens, r_t2v_b if pad else r_t2v, r_v2t, hard_neg, 0.1,
                                             soft, 0.8, 0.7, 0.04, 0.0, True, kd_factor=0.0, **kw)[:2]
            sum(terms).backward()
            out.append(([t.detach().clone() for t in terms], [x.grad.clone() for x in (C, S, P) if x.grad is not None]))
        return out

    ref = run(False, None)
    sw = F_.ScheduleWords(nq_b, nv, soft, DEV)
    for f in (0.1, 0.0):
        sw.words_for(f)
    sw.update(0.8, 0.7, 0.9, nq)
    for got in (run(True, None), run(True, sw)):
        for (t_got, g_got), (t_ref, g_ref) in zip(got, ref):
            # (the sums run over the same terms at other positions of the scratch: the last bits of the three totals may differ)
            assert all(float(a) == pytest.approx(float(b), rel=2e-6) for a, b in zip(t_got, t_ref)), (t_got, t_ref)
            for a, b in zip(g_got, g_ref):
                assert torch.allclose(a[:nq], b, rtol=1e-6, atol=1e-9) and float(a[nq:].abs().max()) == 0.0
    assert sw.used == 2
    # another count through the same words (the next batch of the bucket)
    sw.update(0.8, 0.7, 0.9, nq - 3)
    with pytest.raises(ValueError):
        sw.update(0.8, 0.7, 0.9, nq_b + 1)
