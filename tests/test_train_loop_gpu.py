"""GPU: the training driver end to end on a tiny synthetic dataset: DataLoader + collate_train -> train steps ->
eval_epoch after every epoch -> best checkpoint -> reload.  Captions are noisy copies of a clip of their video
(through a fixed random projection), so the model can learn: SumR must improve and the loss must fall."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class TinySet(torch.utils.data.Dataset):
    def __init__(self, n=24, dv=256, dq=128, seed=0):
        g = torch.Generator().manual_seed(seed)
        proj = torch.randn(dv, dq, generator=g) / dv ** 0.5
        tproj_v = torch.randn(dv, 512, generator=g) / dv ** 0.5
        self.items = []
        for i in range(n):
            L = int(torch.randint(4, 13, (1,), generator=g))
            v = torch.nn.functional.normalize(torch.randn(L, dv, generator=g), dim=-1)
            caps, tcaps = [], []
            for c in range(2):
                l = int(torch.randint(0, L, (1,), generator=g))
                w = v[l] @ proj
                words = torch.nn.functional.normalize(w.unsqueeze(0) + 0.3 * torch.randn(5 + c, dq, generator=g), dim=-1)
                caps.append(words)
                tcaps.append((v[l] @ tproj_v).unsqueeze(0) * 3.0)
            self.items.append((v, caps, v @ tproj_v * 3.0, tcaps, i, [f"v{i}#{c}" for c in range(2)], f"v{i}"))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]

    def videos(self):
        return [(it[0], it[4], it[6]) for it in self.items]

    def texts(self):
        out = []
        for it in self.items:
            for c, cap in enumerate(it[1]):
                out.append((cap, len(out), it[5][c]))
        return out


class L(torch.utils.data.Dataset):
    def __init__(self, x): self.x = x
    def __len__(self): return len(self.x)
    def __getitem__(self, i): return self.x[i]


def test_train_loop_learns_and_checkpoints(tmp_path):
    from dldkd_amd.model import DLDKD
    from dldkd_amd import train as T
    ds = TinySet()
    cfg = types.SimpleNamespace(visual_input_size=256, query_input_size=128, inheritance_hidden=384, exploration_hidden=384,
                                max_ctx_l=128, max_desc_l=30, input_drop=0.1, drop=0.1, n_heads=4, initializer_range=0.02,
                                margin=0.1, use_hard_negative=False, hard_pool_size=5, label_style="soft")
    opt = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                collection="tiny", alpha=0.8, belta=0.8, device=torch.device(DEV), bsz=8, pin_memory=False,
                                num_workers=0, lr=1e-3, wd=0.01, lr_warmup_proportion=0.05, n_epoch=6, max_es_cnt=10,
                                hard_negative_start_epoch=0, hard_pool_size=5, distill_loss_decay="exp", exponential_k=0.95,
                                selfDistil_sigmoid_k=800, alpha_decay="sigmoid", belta_decay="sigmoid", grad_clip=-1,
                                eval_context_bsz=16, eval_query_bsz=50, eval_untrained=True,
                                ckpt_filepath=str(tmp_path / "model.ckpt"))
    torch.manual_seed(0)
    m = DLDKD(cfg, opt)
    hist = T.train(m, ds, L(ds.videos()), L(ds.texts()), opt)
    assert hist[0][0] == -1 and len(hist) == 7                     # eval_untrained epoch + 6 epochs
    sumr = [h[2] for h in hist]
    losses = [h[1]["loss_overall"] for h in hist[1:]]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert max(sumr[1:]) > sumr[0] + 20, sumr                      # it learned to retrieve
    assert m.weight == pytest.approx(0.95 ** 5) and m.config.use_hard_negative is True
    m2, ep = T.load_checkpoint(opt.ckpt_filepath, opt)
    assert 0 <= ep <= 5
    with torch.no_grad():
        from dldkd_amd.eval import eval_epoch
        assert eval_epoch(m2.to(DEV), L(ds.videos()), L(ds.texts()), opt) == pytest.approx(max(sumr))


def _fit(precision, tmp_path, force_ddp=False):
    from dldkd_amd.model import DLDKD
    from dldkd_amd import train as T, ops
    ds = TinySet()
    cfg = types.SimpleNamespace(visual_input_size=256, query_input_size=128, inheritance_hidden=384, exploration_hidden=384,
                                max_ctx_l=128, max_desc_l=30, input_drop=0.1, drop=0.1, n_heads=4, initializer_range=0.02,
                                margin=0.1, use_hard_negative=False, hard_pool_size=5, label_style="soft")
    opt = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                collection="tiny", alpha=0.8, belta=0.8, device=torch.device(DEV), bsz=8, pin_memory=False,
                                num_workers=0, lr=1e-3, wd=0.01, lr_warmup_proportion=0.05, n_epoch=5, max_es_cnt=10,
                                hard_negative_start_epoch=0, hard_pool_size=5, distill_loss_decay="exp", exponential_k=0.95,
                                selfDistil_sigmoid_k=800, alpha_decay="sigmoid", belta_decay="sigmoid", grad_clip=-1,
                                eval_context_bsz=16, eval_query_bsz=50, eval_untrained=True,
                                ckpt_filepath=str(tmp_path / f"model_{precision}.ckpt"))
    torch.manual_seed(0)
    m = DLDKD(cfg, opt)
    ops.set_gemm_precision(precision)
    try:
        return T.train(m, ds, L(ds.videos()), L(ds.texts()), opt)
    finally:
        ops.set_gemm_precision("fp32")


def test_throughput_mode_training_learns_like_parity_mode(tmp_path):
    """Same data, seeds and schedule with every GEMM on bf16 MFMA: the loss curve tracks the parity-mode curve and
    retrieval improves as much (bf16 operand rounding must not change what is learned)."""
    ref = _fit("fp32", tmp_path)
    got = _fit("bf16", tmp_path)
    l_ref = [h[1]["loss_overall"] for h in ref[1:]]
    l_got = [h[1]["loss_overall"] for h in got[1:]]
    assert all(np.isfinite(l_got)) and l_got[-1] < l_got[0]
    for i, (a, b) in enumerate(zip(l_got, l_ref)):
        # the first epochs track closely (measured 3.876/3.874, 1.079/1.074, 0.348/0.338); later the two runs are
        # different trajectories of the same noisy optimisation (0.25/0.20, 0.16/0.13)
        tol = 0.06 if i < 3 else 0.4
        assert abs(a - b) <= tol * abs(b) + 0.02, (l_got, l_ref)
    s_ref, s_got = [h[2] for h in ref], [h[2] for h in got]
    assert max(s_got[1:]) > s_got[0] + 20 and max(s_got[1:]) >= max(s_ref[1:]) - 25, (s_got, s_ref)


def test_data_parallel_step_on_one_rank_rccl_group(tmp_path):
    """The DDP branch of train_step (flat gradient bucket all-reduced over RCCL) with a one-rank group forced on:
    identical history to the plain run (mean over one rank is the identity)."""
    import os
    import torch.distributed as dist
    from dldkd_amd import train as T
    ref = _fit("fp32", tmp_path)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29591", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=torch.device(DEV))
    old = T.DDP_MIN_WORLD
    T.DDP_MIN_WORLD = 1
    try:
        got = _fit("fp32", tmp_path)
    finally:
        T.DDP_MIN_WORLD = old
        dist.destroy_process_group()
    for a, b in zip(got[1:], ref[1:]):
        # not bitwise: split-K weight gradients and LayerNorm gamma/beta gradients accumulate with fp32 atomics, whose
        # order differs run to run; the difference stays at rounding level over the five epochs
        assert a[1]["loss_overall"] == pytest.approx(b[1]["loss_overall"], rel=2e-3)
