"""Batch assembly for the eval driver: the input contract of the hot path (padding + 0/1 masks).

Restates the two eval collates of the reference (method/data_provider.py:75-86,139-170); dataset file
readers (BigFile / HDF5) are out of scope - any Dataset yielding (feat (len, D) float32, index, id) works.
"""
import contextlib

import torch

HOST_THREADS = 16      # cap of torch's intra-op CPU threads while batches are assembled (None: leave it alone)


@contextlib.contextmanager
def host_threads():
    """Batch assembly is many small CPU tensor ops (pad, stack, compare).  With torch's default of one thread per logical
    core they crawl on many-core hosts (measured on a 256-core box: pad_sequence of a 50-caption batch 10 ms with 256
    threads, 0.3 ms with 16), so the eval / train drivers cap the intra-op thread count while they run and restore it."""
    if HOST_THREADS is None or torch.get_num_threads() <= HOST_THREADS:
        yield
        return
    old = torch.get_num_threads()
    torch.set_num_threads(HOST_THREADS)
    try:
        yield
    finally:
        torch.set_num_threads(old)


def _pad(seqs):
    """Zero-pad to the longest item; mask = 1 on real rows.  pad_sequence + one comparison instead of a Python loop
    of per-item slice assignments (the loop was 0.9 s of a 1.7 s eval_epoch on 2000 videos / 6000 captions)."""
    seqs = [torch.as_tensor(s, dtype=torch.float32) for s in seqs]
    lens = torch.tensor([int(s.shape[0]) for s in seqs])
    feats = torch.nn.utils.rnn.pad_sequence(seqs, batch_first=True)
    mask = (torch.arange(int(lens.max()))[None, :] < lens[:, None]).float()
    return feats, mask


def collate_frame_val(data):
    """[(clip feats, idx, video_id)] -> (videos (B, Lmax, Dv), mask (B, Lmax), idxs, video_ids)."""
    feats, idxs, vids = zip(*data)
    videos, mask = _pad(feats)
    return videos, mask, idxs, vids


def collate_text_val(data):
    """[(word feats, idx, cap_id)] -> (words, mask, idxs, cap_ids), batch sorted by length, longest first
    (data_provider.py:153-154: the score-matrix rows follow THIS order)."""
    data = sorted(data, key=lambda x: int(x[0].shape[0]), reverse=True)
    feats, idxs, caps = zip(*data)
    words, mask = _pad(feats)
    return words, mask, idxs, caps


def collate_train(data):
    """Training batch assembly (data_provider.py:111-136): items are
    (student clip feats (len_v, Dv), [caption word feats (len_q, Dq), ...], teacher clip feats (len_v, 512),
     [teacher caption feats (1, 512), ...], idx, cap_ids, video_id).
    Videos are sorted by their number of captions, most first (:116-117: clip_nce_soft splits hard/soft parts by
    row position); text_labels[q] = index of the query's video in the batch."""
    data = sorted(data, key=lambda x: len(x[1]), reverse=True)
    s_vid, caps, t_vid, t_caps, _idxs, _cap_ids, _vids = zip(*data)
    student_videos, student_mask = _pad(s_vid)
    teacher_videos, _ = _pad(t_vid)
    words, labels, t_words = [], [], []
    for vi, (cs, tcs) in enumerate(zip(caps, t_caps)):
        for c, tc in zip(cs, tcs):
            words.append(c)
            t_words.append(tc)
            labels.append(vi)
    student_text, student_text_mask = _pad(words)
    teacher_text, _ = _pad(t_words)
    return dict(student_videos=student_videos, teacher_videos=teacher_videos, student_videos_mask=student_mask,
                student_text=student_text, student_text_mask=student_text_mask, teacher_text=teacher_text,
                text_labels=labels)
