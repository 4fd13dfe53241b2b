"""Feature ingest: the reference's on-disk formats, read in bulk, and the per-video
down-sampling / normalisation / padding done by ONE GPU kernel per batch instead of Python loops per clip.

On-disk formats (reference utils/basic_utils.py:9-68,231-236):
  <dir>/shape.txt   "nr_of_rows ndims"
  <dir>/id.txt      whitespace-separated row ids (ISO-8859-1)
  <dir>/feature.bin row-major float32, one row per frame / clip
  video2frames.txt  a Python-literal dict {video_id: [frame ids]} (the reference eval()s it; we literal_eval)
"""
import ast
import os

import numpy as np
import torch

from . import native

L2_EPS = 1e-5      # data_provider.py:71


class BigFile:
    """Same files as the reference's BigFile; rows come back as one float32 array via a memory map instead of
    a Python list per row."""

    def __init__(self, datadir):
        with open(os.path.join(datadir, "shape.txt")) as f:
            self.nr_of_images, self.ndims = map(int, f.readline().split())
        with open(os.path.join(datadir, "id.txt"), "rb") as f:
            self.names = [str(x, encoding="ISO-8859-1") for x in f.read().strip().split()]
        assert len(self.names) == self.nr_of_images
        self.name2index = dict(zip(self.names, range(self.nr_of_images)))
        self.binary_file = os.path.join(datadir, "feature.bin")
        self._mm = np.memmap(self.binary_file, dtype=np.float32, mode="r", shape=(self.nr_of_images, self.ndims))

    def shape(self):
        return [self.nr_of_images, self.ndims]

    def rows(self, names):
        """(len(names), ndims) float32, in the order given (KeyError on an unknown id)."""
        idx = np.fromiter((self.name2index[n] for n in names), dtype=np.int64, count=len(names))
        return np.asarray(self._mm[idx])

    def read_one(self, name):
        return self._mm[self.name2index[name]].tolist()

    def read(self, requested, isname=True):
        """Reference-compatible: (names, vectors) sorted by row index, unknown names skipped."""
        req = set(requested)
        pairs = sorted(((self.name2index[x], x) for x in req if x in self.name2index) if isname
                       else ((x, self.names[x]) for x in req))
        if not pairs:
            return [], []
        return [p[1] for p in pairs], [self._mm[p[0]].tolist() for p in pairs]


def read_dict(path):
    """video2frames.txt: a Python-literal dict (basic_utils.py:231-236 uses eval)."""
    with open(path) as f:
        return ast.literal_eval(f.read())


def sampling_bounds(num_clips, max_len):
    """(start, end) of every output clip, exactly as uniform_feature_sampling computes them
    (data_provider.py:57-59: numpy round-half-to-even, clipped to num_clips - 1)."""
    if max_len is None or num_clips <= max_len:
        s = np.arange(num_clips, dtype=np.int32)
        return s, s.copy()                       # empty ranges -> "take frame s"
    idxs = np.round(np.arange(0, max_len + 1, 1.0) / max_len * num_clips).astype(np.int32)
    idxs[idxs > num_clips - 1] = num_clips - 1
    return idxs[:-1].copy(), idxs[1:].copy()


def build_video_batch(frame_arrays, max_ctx_l, device):
    """frame_arrays: list of (n_frames_i, D) float32 arrays (raw, un-normalised).  Returns
    (videos (B, Lmax, D) fp32 on `device`, mask (B, Lmax)): down-sampled to <= max_ctx_l clips by segment means,
    L2-normalised per clip, zero padded - the tensors collate_frame_val would have produced."""
    D = frame_arrays[0].shape[1]
    starts, ends, lens, off = [], [], [], 0
    for a in frame_arrays:
        s, e = sampling_bounds(a.shape[0], max_ctx_l)
        starts.append(s + off)
        ends.append(e + off)
        lens.append(len(s))
        off += a.shape[0]
    B, Lmax = len(frame_arrays), max(lens)
    seg_s = np.full((B, Lmax), -1, np.int32)
    seg_e = np.full((B, Lmax), -1, np.int32)
    for i, (s, e) in enumerate(zip(starts, ends)):
        seg_s[i, :len(s)] = s
        seg_e[i, :len(e)] = e
    frames = torch.from_numpy(np.concatenate(frame_arrays, 0)).to(device, non_blocking=True)
    ts, te = torch.from_numpy(seg_s).to(device), torch.from_numpy(seg_e).to(device)
    out = torch.empty(B, Lmax, D, dtype=torch.float32, device=device)
    native.check(native.lib().dldkd_segment_mean_l2norm_f32(native.ptr(frames), native.ptr(ts), native.ptr(te), native.ptr(out),
                                                            B * Lmax, D, L2_EPS, native.stream()), "segment_mean_l2norm")
    mask = torch.from_numpy((seg_s >= 0).astype(np.float32)).to(device)
    return out, mask


def load_gallery_batch(bigfile, video2frames, video_ids, max_ctx_l, device):
    """Bulk counterpart of VisDataSet4DLDKD.__getitem__ + collate_frame_val for a batch of videos."""
    arrays = [bigfile.rows(video2frames[v]) for v in video_ids]
    return build_video_batch(arrays, max_ctx_l, device)
