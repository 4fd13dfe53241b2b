"""GPU: edge cases of the drop-in surface (what a caller of the reference could legally pass)."""
import types

import numpy as np
import pytest
import torch

import dldkd_oracle as orc
import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mk(dv=1024, dq=1024, double_branch=True, label_style="soft", drop=0.0, seed=3):
    from dldkd_amd.model import DLDKD
    cfg = types.SimpleNamespace(visual_input_size=dv, query_input_size=dq, inheritance_hidden=384, exploration_hidden=384,
                                max_ctx_l=128, max_desc_l=30, input_drop=drop, drop=drop, n_heads=4, initializer_range=0.02,
                                margin=0.2, use_hard_negative=False, hard_pool_size=20, label_style=label_style)
    opt = types.SimpleNamespace(double_branch=double_branch, kl_intra_weight=0.1, inher_nce_weight=0.04,
                                explore_nce_weight=0.04, collection="activitynet", alpha=0.8, belta=0.8)
    torch.manual_seed(seed)
    return DLDKD(cfg, opt).to(DEV)


def test_config_may_be_a_dict_or_attribute_object():
    from dldkd_amd.model import DLDKD
    cfg = dict(visual_input_size=1024, query_input_size=1024, inheritance_hidden=384, exploration_hidden=384, max_ctx_l=128,
               max_desc_l=30, input_drop=0.1, drop=0.1, n_heads=4, initializer_range=0.02, margin=0.2,
               use_hard_negative=False, hard_pool_size=20)
    opt = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                collection="x", alpha=0.8, belta=0.8, label_style="hard")
    m = DLDKD(cfg, opt)
    assert m.label_style == "hard"            # falls back to opt when config lacks it (reference quirk)
    m.set_hard_negative(True, 7)
    assert cfg["use_hard_negative"] is True and cfg["hard_pool_size"] == 7
    with pytest.raises(ValueError):
        bad = dict(cfg, n_heads=5)
        DLDKD(bad, opt)


def test_single_branch_model_eval_and_train():
    m = _mk(double_branch=False)
    assert len(m.state_dict()) == 37
    batch = synth.make_train_batch(4, nv=6, caps=2, L=9, dv=1024, dq=1024)
    batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    m.eval()
    with torch.no_grad():
        gi, ge = m.encode_context(batch["student_videos"], batch["student_videos_mask"])
        qi, qe = m.encode_query(batch["student_text"], batch["student_text_mask"])
    assert ge is None and qe is None and gi.shape == (6, 9, 384) and qi.shape == (12, 384)
    fused, s0, s1 = m.pooled_scores([qi], [gi], batch["student_videos_mask"])
    assert s1 is None and torch.equal(fused, s0)
    p = {k: v.cpu() for k, v in m.state_dict().items()}
    ref = orc.sim_scores(orc.encode_query(p, batch["student_text"].cpu(), batch["student_text_mask"].cpu(), double_branch=False)[0],
                         orc.encode_context(p, batch["student_videos"].cpu(), batch["student_videos_mask"].cpu(), double_branch=False)[0],
                         batch["student_videos_mask"].cpu())[0]
    assert (s0.cpu() - ref).abs().max() < 6e-3
    m.train()
    loss, d = m(batch)
    loss.backward()
    assert d["explore_trip"] == 0 and d["explore_nce"] == 0 and torch.isfinite(loss)
    assert all(p_.grad is not None for p_ in m.parameters())


def test_batch_of_one_query_and_one_video():
    m = _mk().eval()
    vid = torch.nn.functional.normalize(torch.randn(1, 5, 1024), dim=-1).to(DEV)
    words = torch.nn.functional.normalize(torch.randn(1, 7, 1024), dim=-1).to(DEV)
    with torch.no_grad():
        gi, ge = m.encode_context(vid, torch.ones(1, 5, device=DEV))
        qi, qe = m.encode_query(words, torch.ones(1, 7, device=DEV))
        qi2, _ = m.encode_query(words[0], torch.ones(7, device=DEV))          # the reference's .squeeze() shape
    assert qi.shape == (1, 384) and torch.allclose(qi, qi2)
    pooled, clip = m.get_sim_scores(qi, gi)                                    # mask=None
    assert pooled.shape == (1, 1) and clip.shape == (1, 5, 1)
    ref = orc.sim_scores(qi.cpu(), gi.cpu())[0]
    assert (pooled.cpu() - ref).abs().max() < 1e-5


def test_maximum_lengths_128_clips_30_words():
    m = _mk().eval()
    vid = torch.nn.functional.normalize(torch.randn(3, 128, 1024), dim=-1).to(DEV)
    words = torch.nn.functional.normalize(torch.randn(4, 30, 1024), dim=-1).to(DEV)
    vm, wm = torch.ones(3, 128, device=DEV), torch.ones(4, 30, device=DEV)
    vm[1, 100:] = 0; wm[2, 11:] = 0
    with torch.no_grad():
        gi, ge = m.encode_context(vid * vm.unsqueeze(-1), vm)
        qi, qe = m.encode_query(words * wm.unsqueeze(-1), wm)
    p = {k: v.cpu() for k, v in m.state_dict().items()}
    ogi, oge = orc.encode_context(p, (vid * vm.unsqueeze(-1)).cpu(), vm.cpu())
    oqi, oqe = orc.encode_query(p, (words * wm.unsqueeze(-1)).cpu(), wm.cpu())
    for a, b in ((gi, ogi), (ge, oge), (qi, oqi), (qe, oqe)):
        assert (a.cpu() - b).abs().max() < 3e-5
    with pytest.raises(Exception):
        m.encode_context(torch.zeros(1, 129, 1024, device=DEV), torch.ones(1, 129, device=DEV))   # > max_ctx_l positions


def test_train_mode_with_dropout_runs_and_is_stochastic():
    m = _mk(drop=0.2).train()
    batch = synth.make_train_batch(6, nv=8, caps=2, L=10, dv=1024, dq=1024)
    batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    torch.manual_seed(1); l1, _ = m(batch)
    torch.manual_seed(2); l2, _ = m(batch)
    assert torch.isfinite(l1) and torch.isfinite(l2) and float(l1) != float(l2)
    l1.backward()
    m.eval()
    torch.manual_seed(1); e1, _ = m(batch)
    for _ in range(6):                        # the forward pass is bitwise reproducible (no atomics on it: split-K GEMMs
        torch.manual_seed(1); e2, _ = m(batch)   # are reserved for the backward layouts), same triplet draws
        assert float(e1) == float(e2)
    from dldkd_amd import ops
    ops.set_gemm_precision("bf16")
    try:
        torch.manual_seed(1); b1, _ = m(batch)
        for _ in range(4):
            torch.manual_seed(1); b2, _ = m(batch)
            assert float(b1) == float(b2)
    finally:
        ops.set_gemm_precision("fp32")


def test_compute_kl_loss_public_method(golden_dir):
    g = np.load(f"{golden_dir}/g3_losses.npz")
    counts = list(g["counts"]); labels = [i for i, c in enumerate(counts) for _ in range(c)]
    L = g["predict"].shape[1]
    mask = torch.from_numpy((np.arange(L)[None] < g["lens"][:, None]).astype(np.float32)).to(DEV)
    m = _mk()
    kl = m.compute_kl_loss(torch.from_numpy(g["predict"]).to(DEV), torch.from_numpy(g["target"]).to(DEV), mask, 0.2,
                           mode="frame_score", query_labels=labels)
    assert abs(float(kl) - float(g["kl"])) <= 1e-4 * abs(float(g["kl"]))
    with pytest.raises(NotImplementedError):
        m.compute_kl_loss(None, None, None, 0.2, mode="batch_score")


def test_checkpoint_round_trip(tmp_path):
    """{"model", "model_cfg", "epoch"} checkpoints (train.py:234) load both ways."""
    m = _mk()
    path = tmp_path / "model.ckpt"
    torch.save({"model": m.state_dict(), "model_cfg": m.config, "epoch": 3}, path)
    ck = torch.load(path, weights_only=False)
    from dldkd_amd.model import DLDKD
    opt = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                collection="activitynet", alpha=0.8, belta=0.8)
    m2 = DLDKD(ck["model_cfg"], opt)
    m2.load_state_dict(ck["model"])
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1.cpu(), v2.cpu())


def test_single_branch_model_throughput_paths_and_stepper():
    """double_branch = False through the round-3 paths: the fused gallery encode (K4 round-1 kernel + the tower kernel with one
    branch) and the fused query tower against the parity-mode towers; the graph stepper with two tower streams and two
    gradient-bucket candidates against the eager step."""
    from dldkd_amd import ops, scoring
    from dldkd_amd import train as T
    from dldkd_amd.optimization import BertAdam
    m = _mk(double_branch=False, drop=0.1)
    assert len(m.grad_buckets()) == 2
    batch = synth.make_train_batch(14, nv=40, caps=2, L=64, len_lo=5, dv=1024, dq=1024)
    batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    vid, vmask = batch["student_videos"], batch["student_videos_mask"]
    m.eval()
    with torch.no_grad():
        gi, _ = m.encode_context(vid, vmask)
        qi, _ = m.encode_query(batch["student_text"], batch["student_text_mask"])
        ref = m.pooled_scores([qi], [gi], vmask)[0]
        ops.set_gemm_precision("bf16")
        m.fast_input_proj = True
        try:
            pk = scoring.GalleryPacker(vid.shape[0], 64, 1, torch.device(DEV))
            lens_host = (vmask > 0).sum(1).cpu().numpy()
            assert m.encode_context_into(pk, vid, vmask, lens_host=lens_host)
            qf, qn = m.encode_query(batch["student_text"], batch["student_text_mask"])
            assert qn is None
            got = m.pooled_scores([qf], pk.finish())[0]
        finally:
            ops.set_gemm_precision("fp32")
            m.fast_input_proj = False
    assert (got - ref).abs().max().item() < 3e-2
    # training: stepper (two tower streams) == eager, step by step
    topt = types.SimpleNamespace(grad_clip=-1)

    def make():
        mm = _mk(double_branch=False, drop=0.1).train()
        return mm, BertAdam([{"params": list(mm.parameters()), "weight_decay": 0.01}], lr=1e-3, warmup=0.1, t_total=50)
    me, oe = make()
    mg, og = make()
    stepper = T.GraphedTrainStep(mg, og, topt)
    for it in range(6):
        og.fp.flat.copy_(oe.fp.flat); og.m.copy_(oe.m); og.v.copy_(oe.v); og.step_count = oe.step_count
        torch.manual_seed(40 + it)
        le, de = T.train_step(me, stepper._bucketed(batch), oe, topt)
        torch.manual_seed(40 + it)
        lg, dg = stepper(batch)
        assert float(le.detach()) == pytest.approx(float(lg), rel=1e-5)
        assert dg["explore_trip"] == 0 and dg["explore_nce"] == 0
        assert (oe.fp.flat - og.fp.flat).abs().max().item() <= 2e-7 + 0.02 * og.get_lr()[0]
    assert stepper.replays == 5
