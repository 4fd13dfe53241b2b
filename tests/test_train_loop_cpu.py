"""CPU: the host logic of the training driver - epoch schedules against independently written closed forms
(reference method/train.py:73-125 with the do_tvr.sh settings), collate_train's ordering contract."""
import math
import types

import torch


def _opt(**kw):
    d = dict(distill_loss_decay="exp", exponential_k=0.95, linear_k=-0.01, linear_b=1.0, sigmoid_k=10.0,
             selfDistil_sigmoid_k=800, alpha=0.8, belta=0.8, alpha_decay="sigmoid", belta_decay="sigmoid", n_epoch=120)
    d.update(kw)
    return types.SimpleNamespace(**d)


def test_epoch_schedules_closed_forms():
    from dldkd_amd.train import epoch_schedules
    for e in (0, 1, 7, 60, 119):
        w, a, b = epoch_schedules(_opt(), e)
        assert w == 0.95 ** e
        s = 800 / (800 + math.exp(e * 100 / 800))
        assert a == max(0.8 * s, 0.0) and b == max(0.8 * s, 0.5)            # belta floor 0.5 since 0.8 >= 0.5
    w, a, b = epoch_schedules(_opt(distill_loss_decay="sigmoid", alpha_decay="cosine", belta_decay="linear", belta=0.3), 30)
    assert w == 10.0 / (10.0 + math.exp(30 * 100 / 10.0))
    assert a == 0.5 * 0.8 * (1 + math.cos(math.pi * 30 / 120))
    assert b == max(0.3 + ((0 - 0.3) / 120) * 30, 0)
    w, a, b = epoch_schedules(_opt(distill_loss_decay="linear", alpha_decay="exp", belta_decay="None"), 200)
    assert w == 0.05 and a == 0.8 * 0.95 ** 200 and b == 0.8
    w, a, b = epoch_schedules(_opt(distill_loss_decay=None, alpha_decay=None, belta_decay=None), 3)
    assert w is None and a is None and b is None


def test_collate_train_contract():
    from dldkd_amd.data import collate_train
    g = torch.Generator().manual_seed(0)

    def item(n_clips, n_caps, vid):
        return (torch.randn(n_clips, 8, generator=g), [torch.randn(3 + i, 6, generator=g) for i in range(n_caps)],
                torch.randn(n_clips, 512, generator=g), [torch.randn(1, 512, generator=g) for _ in range(n_caps)], 0,
                [f"{vid}#{i}" for i in range(n_caps)], vid)
    batch = collate_train([item(5, 1, "a"), item(9, 3, "b"), item(2, 2, "c")])
    assert batch["student_videos"].shape == (3, 9, 8) and batch["teacher_videos"].shape == (3, 9, 512)
    assert batch["text_labels"] == [0, 0, 0, 1, 1, 2]                        # sorted by #captions, most first
    assert batch["student_videos_mask"].sum(1).tolist() == [9, 2, 5]
    assert batch["student_text"].shape == (6, 5, 6) and batch["teacher_text"].shape == (6, 1, 512)
    assert batch["student_text_mask"].sum(1).tolist() == [3, 4, 5, 3, 4, 3]


def test_flat_params_gather_subset_then_rebind():
    """optimization.FlatParams.gather_subset (one tower's gradients copied into the flat buffer by that tower's own graph,
    train.GraphedTrainStep._capture_parallel) followed by rebind_grads: every gradient lands in its flat view once, parameters
    without a gradient are zeroed and flagged, nothing is gathered twice, drop_grads forgets the subsets."""
    from dldkd_amd.optimization import FlatParams
    g = torch.Generator().manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in ((3, 5), (7,), (2, 2, 2), (300,), (4, 4))]
    fp = FlatParams(ps)
    fp.drop_grads()
    grads = [torch.randn(p.shape, generator=g) for p in ps]
    for i in (0, 1, 3):
        ps[i].grad = grads[i].clone()
    fp.grad.fill_(9.0)
    fp.gather_subset([ps[1], ps[3]])                       # "tower A": both have gradients
    assert ps[1].grad.data_ptr() == fp.views()[1].data_ptr() and torch.equal(ps[3].grad, grads[3])
    ps[3].grad.add_(1.0)                                   # a later write through the view must survive the final rebind
    fp.gather_subset([ps[2], ps[3]])                       # "tower B": ps[2] has none; ps[3] is not gathered again
    had = fp.rebind_grads()
    assert had == (True, True, False, True, False)
    for i, p in enumerate(ps):
        assert p.grad.data_ptr() == fp.views()[i].data_ptr()
    assert torch.equal(ps[0].grad, grads[0]) and torch.equal(ps[1].grad, grads[1]) and torch.equal(ps[3].grad, grads[3] + 1.0)
    assert float(ps[2].grad.abs().sum()) == 0.0 and float(ps[4].grad.abs().sum()) == 0.0
    fp.drop_grads()
    assert fp._had_subset == {} and all(p.grad is None for p in ps)


def test_resident_gallery_chunk_plan():
    """eval.ResidentGallery.plan: chunks of whole videos of at most RESIDENT_CHUNK_ROWS clips (a longer video alone), rows
    relative to the chunk, lengths and slot tables per chunk - on a stand-in table (the plan is host logic)."""
    import numpy as np
    from dldkd_amd import eval as ev, ops
    lens = [5, 0, 128, 33, 64, 1, 127, 96, 2]
    res = ev.ResidentGallery.__new__(ev.ResidentGallery)
    res.table = types.SimpleNamespace(lens=lens)
    old = ev.RESIDENT_CHUNK_ROWS
    try:
        ev.RESIDENT_CHUNK_ROWS = 200
        res.plan(torch.device("cpu"))
    finally:
        ev.RESIDENT_CHUNK_ROWS = old
    start = np.concatenate([[0], np.cumsum(lens)])
    assert [c[0] for c in res.chunks] == [0, 4, 7] and sum(c[1] for c in res.chunks) == len(lens)
    for va, n, r0, r1, lens_d, row0_d, items in res.chunks:
        assert (r0, r1) == (start[va], start[va + n]) and r1 - r0 <= 200
        assert lens_d.tolist() == lens[va:va + n] and row0_d.tolist() == (start[va:va + n] - start[va]).tolist()
        want = ops.plan_tower_items(np.asarray(lens[va:va + n]))
        assert items.dtype == torch.int32 and np.array_equal(items.numpy(), want)
        seqs = {int(e) >> 10 for e in want.reshape(-1) if e >= 0}
        assert seqs == {i for i in range(n) if lens[va + i] > 0}          # every non-empty video has its slots, empty ones none
    assert res.lens_dev.tolist() == lens
