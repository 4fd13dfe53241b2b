"""CPU: the oracle (oracle/dldkd_oracle.py) against the committed outputs of the reference.

The golden files were produced by tests/golden/make_golden.py, which ran the upstream reference
itself; inputs are rebuilt here from the same seeds (tests/golden/synth.py).  Tolerances are fp32
rounding of re-associated sums (the oracle is vectorised where the reference loops).
"""
import os

import numpy as np
import pytest
import torch

import dldkd_oracle as orc
import synth

TOL = 2e-6


def _close(a, b, tol=TOL):
    a = np.asarray(torch.as_tensor(a).detach().double())
    b = np.asarray(b, dtype=np.float64)
    scale = max(1.0, np.abs(b).max())
    assert np.abs(a - b).max() <= tol * scale, (np.abs(a - b).max(), scale)


def test_g1_simpool(golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_simpool.npz"))
    rs = np.random.RandomState(11)
    q = torch.from_numpy(rs.standard_normal((7, 384)).astype(np.float32))
    ctx = torch.from_numpy(rs.standard_normal((5, 9, 384)).astype(np.float32))
    mask = torch.from_numpy((np.arange(9)[None] < g["lens"][:, None]).astype(np.float32))
    ctx = ctx * mask.unsqueeze(-1)
    pooled, clip, idx = orc.sim_scores(q, ctx, mask)
    _close(pooled, g["pooled"]); _close(clip, g["clip"])
    assert (idx.numpy() == g["argmax"]).all()
    _close(orc.unnormalized_sim_scores(q, ctx, mask), g["raw"])
    _close(orc.sim_scores(q, ctx)[0], g["pooled_nomask"])
    # masked clips are exactly -1e10 (mask_logits, model.py:444-445)
    assert (clip[:, 3:, 1] == -1e10).all()


@pytest.mark.parametrize("tag,dv,dq,seed", [("tvr", 3072, 768, 21), ("anet", 1024, 1024, 22)])
def test_g2_encoders(golden_dir, tag, dv, dq, seed):
    g = np.load(os.path.join(golden_dir, "g2_encoders.npz"))
    p = synth.make_params(seed, dv, dq)
    rs = np.random.RandomState(seed + 100)
    vid, vmask = synth.make_videos(rs, 6, 12, dv, g[f"{tag}_vlens"])
    txt, tmask = synth.make_texts(rs, 5, 30, dq, g[f"{tag}_qlens"])
    vid, vmask, txt, tmask = [torch.from_numpy(a.astype(np.float32)) for a in (vid, vmask, txt, tmask)]
    gi, ge = orc.encode_context(p, vid, vmask)
    qi, qe = orc.encode_query(p, txt, tmask)
    _close(gi, g[f"{tag}_ctx_inh"]); _close(ge, g[f"{tag}_ctx_exp"])
    _close(qi, g[f"{tag}_q_inh"]); _close(qe, g[f"{tag}_q_exp"])


def _g3_inputs(g):
    counts = list(g["counts"])
    labels = [i for i, c in enumerate(counts) for _ in range(c)]
    L = g["predict"].shape[1]
    mask = torch.from_numpy((np.arange(L)[None] < g["lens"][:, None]).astype(np.float32))
    return labels, mask


def test_g3_losses(golden_dir):
    g = np.load(os.path.join(golden_dir, "g3_losses.npz"))
    labels, mask = _g3_inputs(g)
    predict, target, cos, raw, sims = [torch.from_numpy(g[k]) for k in ("predict", "target", "cos", "raw", "sims")]
    _close(orc.kl_frame_score(predict, target, mask, labels), g["kl"])
    for a in (0.0, 0.3, 0.8, 1.0):
        for b in (0.5, 0.8):
            _close(orc.nce_soft(labels, raw, sims, a, b), g[f"nce_soft_a{a}_b{b}"])
            _close(orc.nce_soft(labels, raw, raw, a, b), g[f"nce_self_a{a}_b{b}"])
    _close(orc.nce_hard(labels, raw), g["nce_hard"])
    for hard in (0, 1):
        r_v2t = torch.from_numpy(g[f"trip_hard{hard}_r_v2t"]) if not hard else None
        r_t2v = torch.from_numpy(g[f"trip_hard{hard}_r_t2v"])
        _close(orc.clip_triplet_loss(cos, labels, 0.1, bool(hard), r_v2t, r_t2v), g[f"trip_hard{hard}"])


def test_g3_triplet_rng_order(golden_dir):
    """draw_triplet_randoms consumes torch's CPU RNG in the reference's order (model.py:366-380)."""
    g = np.load(os.path.join(golden_dir, "g3_losses.npz"))
    labels, _ = _g3_inputs(g)
    for hard in (False, True):
        torch.manual_seed(77)
        r_v2t, r_t2v = orc.draw_triplet_randoms(labels, 12, hard, 5)
        assert (r_t2v.numpy() == g[f"trip_hard{int(hard)}_r_t2v"]).all()
        if not hard:
            assert (r_v2t.numpy() == g["trip_hard0_r_v2t"]).all()


@pytest.mark.parametrize("tag,label_style,hard,caps", [("soft_rand", "soft", False, 1), ("soft_hard", "soft", True, 3),
                                                       ("hard_hard", "hard", True, 1)])
def test_g4_forward_and_grads(golden_dir, tag, label_style, hard, caps):
    g = np.load(os.path.join(golden_dir, "g4_forward.npz"))
    p = {k: v.double().requires_grad_(True) for k, v in synth.make_params(41, 3072, 768).items()}
    batch = synth.make_train_batch(1, nv=64, caps=caps, L=16, dv=3072, dq=768, dtype=torch.float64)
    rnd = []
    for i in range(2):
        v2t = torch.from_numpy(g[f"{tag}_r{i}_v2t"]) if not hard else None
        rnd.append((v2t, torch.from_numpy(g[f"{tag}_r{i}_t2v"])))
    cfg = dict(n_heads=4, margin=0.1, use_hard_negative=hard, label_style=label_style, kl_intra_weight=0.1,
               weight=0.95 ** 2, inher_nce_weight=0.04, explore_nce_weight=0.04, alpha=0.8, belta=0.8)
    d = orc.forward_losses(p, batch, cfg, rnd)
    for k in ("inher_trip", "inher_nce", "explore_trip", "explore_nce", "kl_intra", "loss"):
        ref = float(g[f"{tag}_{k}"])
        assert abs(float(d[k]) - ref) <= 2e-5 * max(1.0, abs(ref)), (k, float(d[k]), ref)
    d["loss"].backward()
    gmax = max(float(np.abs(g[f"{tag}_grad/{n}/sample"]).max()) for n in p)
    nmax = max(float(g[f"{tag}_grad/{n}/norm"]) for n in p)
    for n, t in p.items():
        gr = t.grad.reshape(-1)
        idx = np.unique(np.linspace(0, gr.numel() - 1, min(48, gr.numel())).astype(np.int64))
        ref = g[f"{tag}_grad/{n}/sample"].astype(np.float64)
        scale = max(np.abs(ref).max(), 1e-6 * gmax)
        assert np.abs(gr[idx].numpy() - ref).max() <= 5e-4 * scale, n
        assert abs(float(gr.norm()) - float(g[f"{tag}_grad/{n}/norm"])) <= 5e-4 * max(float(g[f"{tag}_grad/{n}/norm"]), 1e-6 * nmax), n


@pytest.mark.parametrize("tag", [c[0] for c in synth.G4T_CASES])
def test_g4t_train_mode_forward_and_grads(golden_dir, tag):
    """The reference's step in model.train() with dropout 0 (golden G4t): the oracle has no train / eval switch - with p = 0 the
    reference's nn.Dropout is the identity - so this pins that the train-mode reference IS the function the oracle restates."""
    g = np.load(os.path.join(golden_dir, "g4t_forward_train.npz"))
    batch, hard, nv, seed = synth.g4t_batch(tag, dtype=torch.float64)
    p = {k: v.double().requires_grad_(True) for k, v in synth.make_params(seed, 3072, 768).items()}
    assert (batch["student_videos_mask"].sum(1).long().numpy() == g[f"{tag}_lens"]).all()
    rnd = []
    for i in range(2):
        v2t = torch.from_numpy(g[f"{tag}_r{i}_v2t"]) if not hard else None
        rnd.append((v2t, torch.from_numpy(g[f"{tag}_r{i}_t2v"])))
    cfg = dict(n_heads=4, margin=0.1, use_hard_negative=hard, label_style="soft", kl_intra_weight=0.1,
               weight=0.95 ** 3, inher_nce_weight=0.04, explore_nce_weight=0.04, alpha=0.8, belta=0.8)
    d = orc.forward_losses(p, batch, cfg, rnd)
    for k in ("inher_trip", "inher_nce", "explore_trip", "explore_nce", "kl_intra", "loss"):
        ref = float(g[f"{tag}_{k}"].reshape(()))
        assert abs(float(d[k]) - ref) <= 2e-5 * max(1.0, abs(ref)), (k, float(d[k]), ref)
    d["loss"].backward()
    gmax = max(float(np.abs(g[f"{tag}_grad/{n}/sample"]).max()) for n in p)
    nmax = max(float(g[f"{tag}_grad/{n}/norm"]) for n in p)
    for n, t in p.items():
        gr = t.grad.reshape(-1)
        idx = np.unique(np.linspace(0, gr.numel() - 1, min(48, gr.numel())).astype(np.int64))
        ref = g[f"{tag}_grad/{n}/sample"].astype(np.float64)
        scale = max(np.abs(ref).max(), 1e-6 * gmax)
        assert np.abs(gr[idx].numpy() - ref).max() <= 5e-4 * scale, n
        assert abs(float(gr.norm()) - float(g[f"{tag}_grad/{n}/norm"])) <= 5e-4 * max(float(g[f"{tag}_grad/{n}/norm"]), 1e-6 * nmax), n


def test_g5_eval_metrics(golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_eval_epoch.npz"))
    m = orc.eval_metrics(g["inh"], g["exp"], list(g["video_metas"]), list(g["query_metas"]))
    np.testing.assert_allclose(m["inher"], g["perf_inher"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(m["explore"], g["perf_explore"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(m["fused"], g["perf_fused"], rtol=0, atol=1e-9)
    assert abs(m["map"] - float(g["map"])) < 1e-9 and abs(m["sumr"] - float(g["sumr"])) < 1e-9


def test_g6_bert_adam(golden_dir):
    g = np.load(os.path.join(golden_dir, "g6_bert_adam.npz"))
    rs = np.random.RandomState(61)
    shapes = [(384, 16), (384,), (7,)]
    names = ["a.weight", "a.bias", "b.LayerNorm.weight"]
    prm = [torch.from_numpy(rs.standard_normal(s).astype(np.float32)).double() for s in shapes]
    m = [torch.zeros_like(q) for q in prm]
    v = [torch.zeros_like(q) for q in prm]
    for step in range(4):
        grads = [torch.from_numpy((rs.standard_normal(s) * (3.0 if step % 2 else 0.01)).astype(np.float32)).double()
                 for s in shapes]
        for i in range(3):
            prm[i], m[i], v[i] = orc.bert_adam_step(prm[i], grads[i], m[i], v[i], step, 3e-4,
                                                    0.01 if i == 0 else 0.0, 200, 0.01)
            _close(prm[i], g[f"step{step}_{names[i]}"], 1e-6)
