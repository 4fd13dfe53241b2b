"""Batch assembly for the eval driver: the input contract of the hot path (padding + 0/1 masks).

Restates the two eval collates of the reference (method/data_provider.py:75-86,139-170); dataset file
readers (BigFile / HDF5) are out of scope - any Dataset yielding (feat (len, D) float32, index, id) works.
"""
import torch


def _pad(seqs):
    n = max(int(s.shape[0]) for s in seqs)
    feats = torch.zeros(len(seqs), n, seqs[0].shape[-1])
    mask = torch.zeros(len(seqs), n)
    for i, s in enumerate(seqs):
        feats[i, :s.shape[0]] = s
        mask[i, :s.shape[0]] = 1.0
    return feats, mask


def collate_frame_val(data):
    """[(clip feats, idx, video_id)] -> (videos (B, Lmax, Dv), mask (B, Lmax), idxs, video_ids)."""
    feats, idxs, vids = zip(*data)
    videos, mask = _pad(feats)
    return videos, mask, idxs, vids


def collate_text_val(data):
    """[(word feats, idx, cap_id)] -> (words, mask, idxs, cap_ids), batch sorted by length, longest first
    (data_provider.py:153-154: the score-matrix rows follow THIS order)."""
    data = sorted(data, key=lambda x: len(x[0]), reverse=True)
    feats, idxs, caps = zip(*data)
    words, mask = _pad(feats)
    return words, mask, idxs, caps
