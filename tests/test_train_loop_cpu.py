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
