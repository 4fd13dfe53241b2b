"""CPU: host-side pieces of the training loop added in round 6 (no GPU calls): the schedule words' values against the
reference's coefficient rules, the pinned-staging appender, the repeat probe of make_train_loader."""
import math
import types

import numpy as np
import pytest
import torch


def test_schedule_words_values_follow_clip_nce_soft():
    """functional.ScheduleWords.write: cq / cv and the five words per KL factor for (alpha, belta, KD weight, valid queries) are what
    clip_nce_soft computes from them (method/model_components.py:126-199: hard = floor(alpha n), alpha / hard and (1 - alpha) / soft
    weights, zero where a part is empty) - written into a host tensor laid out like the step's staged slot."""
    from dldkd_amd import functional as F_
    nq, nv = 96, 40
    sw = F_.ScheduleWords(nq, nv, True, "cpu", store=torch.zeros(F_.ScheduleWords.words_needed(nq, nv), dtype=torch.int32))
    w_inh, w_exp = sw.words_for(0.1), sw.words_for(0.0)
    assert sw.words_for(0.1) is w_inh and w_inh.data_ptr() != w_exp.data_ptr()
    with pytest.raises(RuntimeError):
        sw.words_for(0.3)                                  # two branches, two slots
    host = torch.zeros(F_.ScheduleWords.words_needed(nq, nv), dtype=torch.int32)
    for alpha, beta, weight, nqv in ((0.8, 0.8, 1.0, 96), (0.37, 0.6, 0.95 ** 9, 77), (0.0, 0.5, 0.05, 65), (1.0, 0.5, 1.0, 96)):
        sw.write(host, alpha, beta, weight, nqv)
        cq, cv, words = F_.ScheduleWords.split(host, nq, nv)
        hq, hv = math.floor(alpha * nqv), math.floor(alpha * nv)
        sq, sv = nqv - hq, nv - hv
        use_hard, use_soft = hq != 0 and hv != 0, sq != 0 and sv != 0
        exp_q = np.full(nq, (1 - alpha) / sq if use_soft else 0.0, dtype=np.float32)
        exp_q[:hq] = alpha / hq if use_hard else 0.0
        exp_v = np.full(nv, (1 - alpha) / sv if use_soft else 0.0, dtype=np.float32)
        exp_v[:hv] = alpha / hv if use_hard else 0.0
        assert np.array_equal(cq.numpy()[:nqv], exp_q[:nqv]) and np.array_equal(cv.numpy(), exp_v)
        for w, f in ((words[0], 0.1), (words[1], 0.0)):
            assert w[0].item() == hq and w[1].item() == hv and w[4].item() == nqv
            assert w.view(torch.float32)[2].item() == np.float32(beta) and w.view(torch.float32)[3].item() == np.float32(f * weight)
    with pytest.raises(ValueError):
        sw.write(host, 0.8, 0.8, 1.0, 97)
    # hard labels (clip_nce, :216-234): 1 / n weights, every row "hard", belta unused
    hw = F_.ScheduleWords(nq, nv, False, "cpu", store=torch.zeros(F_.ScheduleWords.words_needed(nq, nv), dtype=torch.int32))
    hw.words_for(0.1)
    hw.write(host, 0.8, 0.8, 0.9, 80)
    cq, cv, words = F_.ScheduleWords.split(host, nq, nv)
    assert np.allclose(cq.numpy(), 1.0 / 80) and np.allclose(cv.numpy(), 1.0 / nv)
    assert words[0][0].item() == 80 and words[0][1].item() == nv and words[0].view(torch.float32)[2].item() == 0.0


def test_pinned_appender_keeps_rows_in_order_across_buffer_boundaries():
    from dldkd_amd.data import _PinnedAppender
    g = torch.Generator().manual_seed(0)
    seqs = [torch.randn(int(n), 8, generator=g) for n in (3, 9, 1, 25, 10, 2, 0, 7)]
    seqs[2], seqs[3] = seqs[2].double(), torch.from_numpy(seqs[3].numpy().astype(np.float16))
    chunks = []
    app = _PinnedAppender("cpu", chunks, ring_bytes=4 * 8 * 10)          # ten rows per staging buffer
    for x in seqs:
        app.add(x)
    app.flush(final=True)
    assert torch.equal(torch.cat(chunks, 0), torch.cat([x.float() for x in seqs], 0)) and all(c.shape[0] <= 10 for c in chunks)
    a = _PinnedAppender("cpu", [], ring_bytes=1024)
    a.add(torch.zeros(2, 8))
    with pytest.raises(ValueError):
        a.add(torch.zeros(2, 9))                           # another feature width in the same table


def test_items_repeat_probe_and_host_loader_on_cpu():
    """train._items_repeat: two reads of an item agree (the reference's Dataset4DLDKD) / differ (augmentation); the global generators
    stay where they were; on a CPU device make_train_loader keeps the reference's DataLoader whatever the option says."""
    from torch.utils.data import DataLoader
    from dldkd_amd import train as T

    class DS(torch.utils.data.Dataset):
        def __init__(self, noisy):
            self.noisy = noisy
            g = torch.Generator().manual_seed(1)
            self.items = [(torch.randn(5, 8, generator=g), [torch.randn(3, 4, generator=g)], torch.randn(5, 6, generator=g),
                           [torch.randn(1, 6, generator=g)], i, [f"v{i}#0"], f"v{i}") for i in range(6)]

        def __len__(self):
            return len(self.items)

        def __getitem__(self, i):
            it = list(self.items[i])
            if self.noisy:
                it[0] = it[0] + torch.randn_like(it[0]) * np.random.rand()
            return tuple(it)

    torch.manual_seed(5)
    np.random.seed(5)
    s_t, s_n = torch.get_rng_state(), np.random.get_state()[1].copy()
    assert T._items_repeat(DS(False)) and not T._items_repeat(DS(True))
    assert torch.equal(s_t, torch.get_rng_state()) and np.array_equal(s_n, np.random.get_state()[1])
    opt = types.SimpleNamespace(device="cpu", bsz=4, pin_memory=False, num_workers=0, device_resident_train=True)
    assert isinstance(T.make_train_loader(DS(False), opt, 0, 1), DataLoader)
