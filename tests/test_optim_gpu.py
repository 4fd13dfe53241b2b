"""GPU: fused multi-tensor BertAdam against the reference's own parameter trajectories (golden G6), the
threshold count used by gather-free ranking, and one optimiser step on the real model."""
import types

import numpy as np
import pytest
import torch

import dldkd_oracle as orc
import synth
from test_encoder_gpu import _model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_bert_adam_vs_golden_g6(golden_dir):
    from dldkd_amd.optimization import BertAdam
    g = np.load(f"{golden_dir}/g6_bert_adam.npz")
    rs = np.random.RandomState(61)
    shapes = [(384, 16), (384,), (7,)]
    names = ["a.weight", "a.bias", "b.LayerNorm.weight"]
    prm = [torch.nn.Parameter(torch.from_numpy(rs.standard_normal(s).astype(np.float32)).to(DEV)) for s in shapes]
    groups = [{"params": [prm[0]], "weight_decay": 0.01}, {"params": prm[1:], "weight_decay": 0.0}]
    opt = BertAdam(groups, lr=3e-4, weight_decay=0.01, warmup=0.01, t_total=200, schedule="warmup_linear")
    for step in range(4):
        grads = [torch.from_numpy((rs.standard_normal(s) * (3.0 if step % 2 else 0.01)).astype(np.float32)) for s in shapes]
        for q, gr in zip(prm, grads):
            q.grad = gr.to(DEV)                     # a fresh tensor, like autograd would leave
        opt.step()
        for i in range(3):
            ref = g[f"step{step}_{names[i]}"]
            assert np.abs(prm[i].detach().cpu().numpy() - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), (step, i)
    assert opt.get_lr()[0] == pytest.approx(3e-4 * orc.warmup_linear(4, 200, 0.01))


def test_bert_adam_split_step_vs_golden_g6(golden_dir):
    """The step as the graphed trainer runs it - the norm scratch zeroed, the gradients gathered subset by subset with their sums of
    squares (dldkd_gather_sumsq_f32), then the update alone (dldkd_bert_adam_update_f32) - follows the reference's trajectories like
    the one-call step; the gathered flat gradients are the sources bit for bit (also from an unaligned source)."""
    from dldkd_amd.optimization import BertAdam
    g = np.load(f"{golden_dir}/g6_bert_adam.npz")
    rs = np.random.RandomState(61)
    shapes = [(384, 16), (384,), (7,)]
    names = ["a.weight", "a.bias", "b.LayerNorm.weight"]
    prm = [torch.nn.Parameter(torch.from_numpy(rs.standard_normal(s).astype(np.float32)).to(DEV)) for s in shapes]
    groups = [{"params": [prm[0]], "weight_decay": 0.01}, {"params": prm[1:], "weight_decay": 0.0}]
    opt = BertAdam(groups, lr=3e-4, weight_decay=0.01, warmup=0.01, t_total=200, schedule="warmup_linear")
    for step in range(4):
        grads = [torch.from_numpy((rs.standard_normal(s) * (3.0 if step % 2 else 0.01)).astype(np.float32)) for s in shapes]
        opt.zero_grad()
        pad = torch.zeros(384 + 1, device=DEV)
        pad[1:] = grads[1].to(DEV)
        for q, gr in zip(prm, grads):
            q.grad = gr.to(DEV)
        prm[1].grad = pad[1:]                       # a source that is not 16-byte aligned
        opt.host_prepare()
        opt.zero_norms()
        opt.fp.gather_subset([prm[0], prm[2]], norm2=opt.norm2)
        assert not opt.fp.norms_ready()
        opt.fp.gather_subset([prm[1]], norm2=opt.norm2)
        assert opt.fp.norms_ready()
        for q, gr, v in zip(prm, grads, opt.fp.views()):
            assert q.grad is v and torch.equal(v.cpu(), gr)
        n2 = opt.norm2.cpu().numpy()
        for i, gr in enumerate(grads):
            assert n2[i] == pytest.approx(float((gr.double() ** 2).sum()), rel=1e-5)
        opt.enqueue()
        for i in range(3):
            ref = g[f"step{step}_{names[i]}"]
            assert np.abs(prm[i].detach().cpu().numpy() - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), (step, i)


def test_gather_sumsq_many_tensors_and_sizes():
    """dldkd_gather_sumsq_f32 over > 32 tensors of awkward sizes (1, 3, 4095, 4096, 4097 ... elements): flat ranges equal the
    sources, untouched padding stays zero, norms match fp64."""
    from dldkd_amd.optimization import FlatParams
    gen = torch.Generator().manual_seed(5)
    sizes = [1, 3, 4, 255, 256, 257, 4095, 4096, 4097, 12289, 384 * 384] + [5 + 7 * i for i in range(30)]
    prm = [torch.nn.Parameter(torch.zeros(n, device=DEV)) for n in sizes]
    fp = FlatParams(prm)
    fp.drop_grads()
    grads = [torch.randn(n, generator=gen) for n in sizes]
    for q, gr in zip(prm, grads):
        q.grad = gr.to(DEV)
    norm2 = torch.zeros(len(sizes), device=DEV)
    fp.gather_subset(prm, norm2=norm2)
    torch.cuda.synchronize()
    flat = fp.grad.cpu()
    covered = torch.zeros(flat.numel(), dtype=torch.bool)
    for i, gr in enumerate(grads):
        s0 = fp._starts[i]
        assert torch.equal(flat[s0:s0 + gr.numel()], gr), i
        covered[s0:s0 + gr.numel()] = True
        assert float(norm2[i]) == pytest.approx(float((gr.double() ** 2).sum()), rel=2e-5), i
    assert not flat[~covered].any()


def test_count_above_matches_torch():
    from dldkd_amd import dist as ddist
    g = torch.Generator().manual_seed(1)
    s = torch.randn(33, 1000, generator=g)
    thr = torch.randn(33, generator=g)
    got = ddist._count_above_hip(s.to(DEV), thr.to(DEV), 777).cpu()
    assert (got == (s[:, :777] > thr[:, None]).sum(1).int()).all()


def test_train_step_updates_all_parameters():
    """forward + backward + fused BertAdam on the TVR-dimension model: every parameter moves, loss finite,
    gradients land in the flat buffer the data-parallel all-reduce uses."""
    from dldkd_amd.optimization import BertAdam
    m = _model(3072, 768, synth.make_params(5, 3072, 768))
    m.train()
    m.set_hard_negative(True, 20)
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    named = list(m.named_parameters())
    groups = [{"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
              {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
    opt = BertAdam(groups, lr=3e-4, weight_decay=0.01, warmup=0.01, t_total=100, schedule="warmup_linear")
    before = {n: p.detach().clone() for n, p in named}
    batch = synth.make_train_batch(3, nv=16, caps=2, L=12, dv=3072, dq=768)
    batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    for _ in range(2):                               # step 0 has lr multiplier 0 (warmup_linear), step 1 moves
        opt.zero_grad()
        loss, d = m(batch)
        assert torch.isfinite(loss)
        loss.backward()
        opt.step()
    assert float(opt.fp.grad.abs().sum()) > 0
    moved = [n for n, p in named if not torch.equal(p.detach(), before[n])]
    assert len(moved) == 74


def test_parameters_without_gradient_are_skipped():
    """`if p.grad is None: continue` (reference optimization.py:294-295): a parameter that received no gradient keeps its value
    AND its moments - no weight decay, no moment decay - while the others step; when it gets a gradient again it steps from
    the moments it had."""
    from dldkd_amd.optimization import BertAdam
    torch.manual_seed(0)
    a = torch.nn.Parameter(torch.randn(300, device=DEV))
    b = torch.nn.Parameter(torch.randn(10, 40, device=DEV))
    opt = BertAdam([{"params": [a, b], "weight_decay": 0.01}], lr=1e-2, warmup=-1, t_total=-1, schedule="none")
    a.grad, b.grad = torch.randn(300, device=DEV), torch.randn(10, 40, device=DEV)
    opt.step()
    a1, b1 = a.detach().clone(), b.detach().clone()
    m1 = opt.m.clone()
    opt.zero_grad()
    a.grad = torch.randn(300, device=DEV)            # b gets nothing this step
    opt.step()
    assert not torch.equal(a.detach(), a1)
    assert torch.equal(b.detach(), b1)               # untouched: no weight decay, no lr * m / sqrt(v)
    sb = opt.fp._starts[1]
    assert torch.equal(opt.m[sb:sb + 400], m1[sb:sb + 400])      # moments did not decay
    opt.zero_grad()
    a.grad, b.grad = torch.randn(300, device=DEV), torch.randn(10, 40, device=DEV)
    opt.step()
    assert not torch.equal(b.detach(), b1)


def test_count_above_and_ranks_with_nan_scores():
    """NaN policy (rank.hip): NaN scores count as above, a NaN ground-truth score ranks nv + 1: a diverged model must not
    report R@K = 100 (with s > gt every comparison against NaN is false -> rank 1 for every query)."""
    from dldkd_amd import dist as ddist
    from dldkd_amd import eval as ev
    s = torch.full((4, 200), float("nan"), device=DEV)
    gts = {q: [q] for q in range(4)}
    rb, rf = ev.gt_ranks_gpu(s, gts)
    assert rb.cpu().tolist() == [201] * 4 and rf.cpu().tolist() == [201] * 4
    assert ev.eval_q2m(-s, gts)[:4] == (0.0, 0.0, 0.0, 0.0)
    g = torch.Generator().manual_seed(2)
    s = torch.randn(6, 40, generator=g)
    s[0, 7] = float("nan")                            # a NaN competitor counts as above the ground truth
    s[1, 1] = float("nan")                            # a NaN ground truth ranks last
    rb, _ = ev.gt_ranks_gpu(s.to(DEV), {q: [q] for q in range(6)})
    want0 = 1 + int((s[0] > s[0, 0]).sum()) + 1
    assert rb.cpu().tolist()[0] == want0 and rb.cpu().tolist()[1] == 41
    for q in range(2, 6):
        assert int(rb[q]) == 1 + int((s[q] > s[q, q]).sum())
    thr = torch.tensor([0.0, float("nan")])
    c = ddist._count_above_hip(torch.tensor([[1.0, float("nan"), -1.0], [0.5, 0.1, float("nan")]], device=DEV), thr.to(DEV), 3)
    assert c.cpu().tolist() == [2, 3]
