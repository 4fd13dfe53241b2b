"""Generate the golden vectors by RUNNING THE REFERENCE ITSELF (build container only).

    python tests/golden/make_golden.py        # writes tests/golden/*.npz

The reference (/root/reference, read-only) is imported unmodified through _ref_import.py; inputs
and weights come from synth.py (seeded, so only outputs are stored).  Each block also asserts
that oracle/dldkd_oracle.py reproduces the reference on the spot, so a generator run doubles as
the oracle's pinning run.  Nothing here is needed, or available, on the GPU box.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import dldkd_oracle as orc          # noqa: E402
import synth                        # noqa: E402
from _ref_import import import_reference   # noqa: E402

R = import_reference()
torch.set_num_threads(8)


def ref_model(dv, dq, params, label_style="soft", hard=False, margin=0.1, alpha=0.8, belta=0.8, drop=0.2, train=False):
    cfg = R.EasyDict(visual_input_size=dv, query_input_size=dq, inheritance_hidden=384,
                     exploration_hidden=384, max_ctx_l=128, max_desc_l=30, input_drop=drop, drop=drop,
                     n_heads=4, initializer_range=0.02, device=[0], margin=margin,
                     use_hard_negative=hard, hard_pool_size=20, label_style=label_style)
    opt = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04,
                                explore_nce_weight=0.04, collection="tvr", alpha=alpha, belta=belta)
    m = R.model.DLDKD(cfg, opt)
    m.load_state_dict(params, strict=True)
    m.train(train)
    return m


def close(a, b, tol, what):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    err = (a - b).abs().max().item()
    scale = max(b.abs().max().item(), 1e-30)
    assert err <= tol * max(1.0, scale), f"{what}: max err {err:.3e} (scale {scale:.3e})"
    return err


def sample_idx(n, k=48):
    return np.unique(np.linspace(0, n - 1, min(k, n)).astype(np.int64))


def g1_simpool():
    rs = np.random.RandomState(11)
    q = torch.from_numpy(rs.standard_normal((7, 384)).astype(np.float32))
    ctx = torch.from_numpy(rs.standard_normal((5, 9, 384)).astype(np.float32))
    lens = np.array([9, 3, 1, 6, 8])
    mask = torch.from_numpy((np.arange(9)[None] < lens[:, None]).astype(np.float32))
    ctx = ctx * mask.unsqueeze(-1)
    pooled, clip = R.model.DLDKD.get_sim_scores(q, ctx, mask)
    raw = R.model.DLDKD.get_unnormalized_sim_scores(q, ctx, mask)
    pooled_nomask, clip_nomask = R.model.DLDKD.get_sim_scores(q, ctx)
    o_pooled, o_clip, o_idx = orc.sim_scores(q, ctx, mask)
    close(o_pooled, pooled, 1e-6, "g1 pooled")
    close(o_clip, clip, 1e-6, "g1 clip")
    close(orc.unnormalized_sim_scores(q, ctx, mask), raw, 1e-6, "g1 raw")
    close(orc.sim_scores(q, ctx)[0], pooled_nomask, 1e-6, "g1 nomask")
    np.savez(os.path.join(HERE, "g1_simpool.npz"), lens=lens, pooled=pooled.numpy(), clip=clip.numpy(),
             raw=raw.numpy(), pooled_nomask=pooled_nomask.numpy(),
             argmax=torch.max(clip, dim=1)[1].numpy())


def g2_encoders():
    out = {}
    for tag, dv, dq, seed in (("tvr", 3072, 768, 21), ("anet", 1024, 1024, 22)):
        p = synth.make_params(seed, dv, dq)
        m = ref_model(dv, dq, p)
        rs = np.random.RandomState(seed + 100)
        lens = np.array([12, 5, 1, 9, 12, 7])
        vid, vmask = synth.make_videos(rs, 6, 12, dv, lens)
        qlens = np.array([30, 11, 5, 17, 1])
        txt, tmask = synth.make_texts(rs, 5, 30, dq, qlens)
        vid, vmask, txt, tmask = [torch.from_numpy(a.astype(np.float32)) for a in (vid, vmask, txt, tmask)]
        with torch.no_grad():
            gi, ge = m.encode_context(vid, vmask)
            qi, qe = m.encode_query(txt, tmask)
        ogi, oge = orc.encode_context(p, vid, vmask)
        oqi, oqe = orc.encode_query(p, txt, tmask)
        close(ogi, gi, 2e-6, tag + " ctx inh"); close(oge, ge, 2e-6, tag + " ctx exp")
        close(oqi, qi, 2e-6, tag + " q inh"); close(oqe, qe, 2e-6, tag + " q exp")
        out.update({f"{tag}_vlens": lens, f"{tag}_qlens": qlens, f"{tag}_ctx_inh": gi.numpy(),
                    f"{tag}_ctx_exp": ge.numpy(), f"{tag}_q_inh": qi.numpy(), f"{tag}_q_exp": qe.numpy()})
    np.savez(os.path.join(HERE, "g2_encoders.npz"), **out)


def g3_losses():
    out = {}
    rs = np.random.RandomState(31)
    nv, L = 12, 10
    counts = [3, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1]
    labels = [i for i, c in enumerate(counts) for _ in range(c)]
    nq = len(labels)
    lens = rs.randint(2, L + 1, size=nv); lens[0] = L
    mask = torch.from_numpy((np.arange(L)[None] < lens[:, None]).astype(np.float32))
    predict = torch.from_numpy(rs.uniform(-1, 1, (nq, L, nv)).astype(np.float32))
    target = torch.from_numpy(rs.uniform(-1, 1, (nq, L, nv)).astype(np.float32))
    cos = torch.from_numpy(rs.uniform(-1, 1, (nq, nv)).astype(np.float32))
    raw = torch.from_numpy((4 * rs.standard_normal((nq, nv))).astype(np.float32))
    sims = torch.from_numpy((6 * rs.standard_normal((nq, nv))).astype(np.float32))
    out.update(lens=lens, counts=np.array(counts))
    p = synth.make_params(1, 64, 64)
    m = ref_model(64, 64, p)
    label_dict = {}
    for i, l in enumerate(labels):
        label_dict.setdefault(l, []).append(i)

    kl = m.compute_kl_loss(predict, target, mask, 0.2, mode="frame_score", query_labels=labels)
    close(orc.kl_frame_score(predict, target, mask, labels), kl, 2e-6, "kl")
    out["kl"] = kl.numpy()

    for a in (0.0, 0.3, 0.8, 1.0):
        for b in (0.5, 0.8):
            v = m.nce_criterion_soft(labels, label_dict, raw, sims, a, b)
            v = torch.as_tensor(v).reshape(())
            close(orc.nce_soft(labels, raw, sims, a, b), v, 2e-6, f"nce_soft a={a} b={b}")
            out[f"nce_soft_a{a}_b{b}"] = v.numpy()
            v2 = torch.as_tensor(m.nce_criterion_soft(labels, label_dict, raw, raw, a, b)).reshape(())
            close(orc.nce_soft(labels, raw, raw, a, b), v2, 2e-6, f"nce_soft self a={a} b={b}")
            out[f"nce_self_a{a}_b{b}"] = v2.numpy()
    v = m.nce_criterion(labels, label_dict, raw)
    close(orc.nce_hard(labels, raw), v, 2e-6, "nce_hard")
    out["nce_hard"] = v.numpy()

    for hard in (False, True):
        m.set_hard_negative(hard, 5)
        torch.manual_seed(77)
        v = m.get_clip_triplet_loss(cos, labels)
        torch.manual_seed(77)
        r_v2t, r_t2v = orc.draw_triplet_randoms(labels, nv, hard, 5)
        close(orc.clip_triplet_loss(cos, labels, 0.1, hard, r_v2t, r_t2v), v, 2e-6, f"triplet hard={hard}")
        out[f"trip_hard{int(hard)}"] = v.numpy()
        out[f"trip_hard{int(hard)}_r_t2v"] = r_t2v.numpy()
        if r_v2t is not None:
            out[f"trip_hard{int(hard)}_r_v2t"] = r_v2t.numpy()
    np.savez(os.path.join(HERE, "g3_losses.npz"), predict=predict.numpy(), target=target.numpy(),
             cos=cos.numpy(), raw=raw.numpy(), sims=sims.numpy(), **out)


def g4_forward():
    """Full DLDKD.forward (eval mode = dropout off) + backward at the C1 shape."""
    out = {}
    for tag, label_style, hard, caps in (("soft_rand", "soft", False, 1), ("soft_hard", "soft", True, 3),
                                         ("hard_hard", "hard", True, 1)):
        dv, dq = 3072, 768
        p = synth.make_params(41, dv, dq)
        m = ref_model(dv, dq, p, label_style=label_style, hard=hard)
        m.weight = 0.95 ** 2
        batch = synth.make_train_batch(1, nv=64, caps=caps, L=16, dv=dv, dq=dq)
        labels = batch["text_labels"]
        torch.manual_seed(4242)
        loss, d = m(batch)
        m.zero_grad()
        loss.backward()
        torch.manual_seed(4242)
        r0 = orc.draw_triplet_randoms(labels, 64, hard, 20)
        r1 = orc.draw_triplet_randoms(labels, 64, hard, 20)
        cfg = dict(n_heads=4, margin=0.1, use_hard_negative=hard, label_style=label_style,
                   kl_intra_weight=0.1, weight=0.95 ** 2, inher_nce_weight=0.04, explore_nce_weight=0.04,
                   alpha=0.8, belta=0.8)
        p64 = {k: v.double().requires_grad_(True) for k, v in p.items()}
        b64 = {k: (v.double() if torch.is_tensor(v) else v) for k, v in batch.items()}
        od = orc.forward_losses(p64, b64, cfg, (r0, r1))
        od["loss"].backward()
        for k in ("inher_trip", "inher_nce", "explore_trip", "explore_nce", "kl_intra"):
            close(od[k], torch.as_tensor(d[k]).reshape(()), 2e-5, f"{tag} {k}")
            out[f"{tag}_{k}"] = torch.as_tensor(d[k]).detach().reshape(()).numpy()
        close(od["loss"], loss, 2e-5, f"{tag} loss")
        out[f"{tag}_loss"] = loss.detach().numpy()
        for i, r in enumerate((r0, r1)):
            if r[0] is not None:
                out[f"{tag}_r{i}_v2t"] = r[0].numpy()
            out[f"{tag}_r{i}_t2v"] = r[1].numpy()
        worst = 0.0
        # key biases have mathematically zero gradient (softmax shift invariance); floor the
        # per-tensor scale at 1e-6 of the largest gradient so rounding noise is not "error".
        gmax = max(q.grad.abs().max().item() for q in m.parameters())
        for name, prm in m.named_parameters():
            g = prm.grad.detach().reshape(-1)
            og = p64[name].grad.reshape(-1)
            e = (og - g.double()).abs().max().item() / max(g.abs().max().item(), 1e-6 * gmax)
            worst = max(worst, e)
            idx = sample_idx(g.numel())
            out[f"{tag}_grad/{name}/norm"] = np.float64(g.double().norm().item())
            out[f"{tag}_grad/{name}/sum"] = np.float64(g.double().sum().item())
            out[f"{tag}_grad/{name}/sample"] = g[idx].numpy()
        assert worst < 5e-4, f"{tag}: oracle fp64 grads vs reference fp32 grads rel err {worst}"
        print(f"  g4 {tag}: loss {float(loss):.6f}  worst grad rel err (oracle fp64 vs ref fp32) {worst:.2e}")
    np.savez(os.path.join(HERE, "g4_forward.npz"), **out)


def g4t_forward_train_mode():
    """Full DLDKD.forward + backward with the reference in model.train() and drop = input_drop = 0 (nn.Dropout(0) is the identity:
    the step is deterministic) on a batch whose videos are padded to L = 64 with 3..64 valid clips - about half of them leave a
    whole 32-row group of padding, the rows the build's training towers skip.  7 losses + all 74 gradients."""
    out = {}
    for tag, hard, caps, nv, seed in synth.G4T_CASES:
        dv, dq = 3072, 768
        p = synth.make_params(seed, dv, dq)
        m = ref_model(dv, dq, p, label_style="soft", hard=hard, drop=0.0, train=True)
        assert m.training
        m.weight = 0.95 ** 3
        batch = synth.make_train_batch(seed, nv=nv, caps=synth.g4t_caps(caps, nv), L=64, len_lo=3, dv=dv, dq=dq)
        lens = batch["student_videos_mask"].sum(1).long()
        assert int((lens <= 32).sum()) >= nv // 4, "the batch must hold videos with an all-padding 32-row group"
        labels = batch["text_labels"]
        torch.manual_seed(777)
        loss, d = m(batch)
        m.zero_grad()
        loss.backward()
        torch.manual_seed(777)
        r0 = orc.draw_triplet_randoms(labels, nv, hard, 20)
        r1 = orc.draw_triplet_randoms(labels, nv, hard, 20)
        cfg = dict(n_heads=4, margin=0.1, use_hard_negative=hard, label_style="soft", kl_intra_weight=0.1, weight=0.95 ** 3,
                   inher_nce_weight=0.04, explore_nce_weight=0.04, alpha=0.8, belta=0.8)
        p64 = {k: v.double().requires_grad_(True) for k, v in p.items()}
        b64 = {k: (v.double() if torch.is_tensor(v) else v) for k, v in batch.items()}
        od = orc.forward_losses(p64, b64, cfg, (r0, r1))
        od["loss"].backward()
        for k in ("inher_trip", "inher_nce", "explore_trip", "explore_nce", "kl_intra"):
            close(od[k], torch.as_tensor(d[k]).reshape(()), 2e-5, f"{tag} {k}")
            out[f"{tag}_{k}"] = torch.as_tensor(d[k]).detach().reshape(()).numpy()
        close(od["loss"], loss, 2e-5, f"{tag} loss")
        out[f"{tag}_loss"] = loss.detach().numpy()
        out[f"{tag}_lens"] = lens.numpy()
        for i, r in enumerate((r0, r1)):
            if r[0] is not None:
                out[f"{tag}_r{i}_v2t"] = r[0].numpy()
            out[f"{tag}_r{i}_t2v"] = r[1].numpy()
        worst = 0.0
        gmax = max(q.grad.abs().max().item() for q in m.parameters())
        for name, prm in m.named_parameters():
            g = prm.grad.detach().reshape(-1)
            og = p64[name].grad.reshape(-1)
            e = (og - g.double()).abs().max().item() / max(g.abs().max().item(), 1e-6 * gmax)
            worst = max(worst, e)
            idx = sample_idx(g.numel())
            out[f"{tag}_grad/{name}/norm"] = np.float64(g.double().norm().item())
            out[f"{tag}_grad/{name}/sum"] = np.float64(g.double().sum().item())
            out[f"{tag}_grad/{name}/sample"] = g[idx].numpy()
        assert worst < 5e-4, f"{tag}: oracle fp64 grads vs reference fp32 grads rel err {worst}"
        print(f"  g4t {tag}: loss {float(loss):.6f}  videos with <= 32 clips {int((lens <= 32).sum())}/{nv}  "
              f"worst grad rel err (oracle fp64 vs ref fp32) {worst:.2e}")
    np.savez(os.path.join(HERE, "g4t_forward_train.npz"), **out)


def g5_eval_epoch():
    dv, dq = 3072, 768
    p = synth.make_params(51, dv, dq)
    m = ref_model(dv, dq, p)
    vids, txts = synth.make_eval_sets(5, nv=64, caps=3, dv=dv, dq=dq)
    opt = types.SimpleNamespace(eval_context_bsz=25, eval_query_bsz=50, num_workers=0, pin_memory=False,
                                device=torch.device("cpu"), double_branch=True)
    with torch.no_grad():
        ctx = R.eval.compute_context_info(m, synth.ListDataset(list(vids)), opt)
        inh, exp, _, qmetas = R.eval.compute_query2ctx_info(m, synth.ListDataset(list(txts)), opt, ctx)
        sumr = R.eval.eval_epoch(m, synth.ListDataset(list(vids)), synth.ListDataset(list(txts)), opt)
    _, t2v = R.eval.get_gt(ctx["video_metas"], qmetas)
    fused = 0.7 * inh + 0.3 * exp
    perf = {k: R.eval.eval_q2m(-1 * s, t2v) for k, s in (("inher", inh), ("explore", exp), ("fused", fused))}
    mp = R.eval.t2v_map(-1 * fused, t2v)
    om = orc.eval_metrics(inh, exp, ctx["video_metas"], qmetas)
    for k in perf:
        close(om[k], perf[k], 1e-9, f"g5 {k}")
    close(om["map"], mp, 1e-9, "g5 map")
    close(om["sumr"], sumr, 1e-9, "g5 sumr")
    np.savez(os.path.join(HERE, "g5_eval_epoch.npz"), inh=inh, exp=exp, query_metas=np.array(qmetas),
             video_metas=np.array(ctx["video_metas"]), perf_inher=np.array(perf["inher"]),
             perf_explore=np.array(perf["explore"]), perf_fused=np.array(perf["fused"]), map=mp, sumr=sumr,
             gallery_inh_sample=ctx["inher_frame_feat"][::7, ::3, ::16].numpy(),
             video_mask=ctx["video_mask"].numpy())
    print(f"  g5: sumr {sumr:.3f}  fused {perf['fused']}")


def g6_bert_adam():
    rs = np.random.RandomState(61)
    shapes = [(384, 16), (384,), (7,)]
    names = ["a.weight", "a.bias", "b.LayerNorm.weight"]
    prm = [torch.nn.Parameter(torch.from_numpy(rs.standard_normal(s).astype(np.float32))) for s in shapes]
    groups = [{"params": [prm[0]], "weight_decay": 0.01}, {"params": prm[1:], "weight_decay": 0.0}]
    optim = R.optim.BertAdam(groups, lr=3e-4, weight_decay=0.01, warmup=0.01, t_total=200,
                             schedule="warmup_linear")
    o_p = [q.detach().double().clone() for q in prm]
    o_m = [torch.zeros_like(q) for q in o_p]
    o_v = [torch.zeros_like(q) for q in o_p]
    out = {}
    for step in range(4):
        grads = [torch.from_numpy((rs.standard_normal(s) * (3.0 if step % 2 else 0.01)).astype(np.float32))
                 for s in shapes]
        for q, g in zip(prm, grads):
            q.grad = g.clone()
        optim.step()
        for i in range(3):
            wd = 0.01 if i == 0 else 0.0
            o_p[i], o_m[i], o_v[i] = orc.bert_adam_step(o_p[i], grads[i].double(), o_m[i], o_v[i], step,
                                                        3e-4, wd, 200, 0.01)
            close(o_p[i], prm[i].detach(), 1e-6, f"bertadam step {step} tensor {i}")
            out[f"step{step}_{names[i]}"] = prm[i].detach().numpy().copy()
    np.savez(os.path.join(HERE, "g6_bert_adam.npz"), **out)


def g7_ingest():
    """uniform_feature_sampling + l2_normalize_np_array (data_provider.py:52-73) and BigFile.read_one on a tiny
    on-disk feature file written here (utils/basic_utils.py:9-68)."""
    import tempfile
    from utils.basic_utils import BigFile
    rs = np.random.RandomState(71)
    out = {}
    for n, max_len in ((5, 8), (8, 8), (9, 8), (13, 8), (100, 16), (129, 128), (300, 128), (777, 128), (3, 1)):
        f = rs.standard_normal((n, 12)).astype(np.float32)
        ref = R.data.l2_normalize_np_array(R.data.uniform_feature_sampling(f, max_len))
        mine = orc.l2_normalize_rows(orc.uniform_feature_sampling(f, max_len))
        close(mine, ref, 1e-6, f"ingest n={n} max_len={max_len}")
        out[f"n{n}_L{max_len}"] = ref.astype(np.float32)
    with tempfile.TemporaryDirectory() as d:
        rows = rs.standard_normal((7, 6)).astype(np.float32)
        ids = [f"vid{i}_f{i*3}" for i in range(7)]
        rows.tofile(os.path.join(d, "feature.bin"))
        open(os.path.join(d, "id.txt"), "w").write(" ".join(ids))
        open(os.path.join(d, "shape.txt"), "w").write("7 6")
        bf = BigFile(d)
        got = np.array([bf.read_one(i) for i in (ids[4], ids[0], ids[6])], dtype=np.float32)
        close(got, rows[[4, 0, 6]], 0, "bigfile read_one")
        names, vecs = bf.read([ids[5], ids[1], "missing"])
        assert names == [ids[1], ids[5]]
        out["bigfile_rows"] = rows
        out["bigfile_read_names"] = np.array(names)
        out["bigfile_read_vecs"] = np.array(vecs, dtype=np.float32)
    np.savez(os.path.join(HERE, "g7_ingest.npz"), **out)


if __name__ == "__main__":
    import warnings
    warnings.filterwarnings("ignore")
    fns = (g1_simpool, g2_encoders, g3_losses, g4_forward, g4t_forward_train_mode, g5_eval_epoch, g6_bert_adam, g7_ingest)
    only = set(sys.argv[1:])
    for fn in [f for f in fns if not only or f.__name__ in only]:
        print(fn.__name__)
        fn()
    print("golden vectors written to", HERE)
