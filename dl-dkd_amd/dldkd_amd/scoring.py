"""Host side of the scoring path (K1 `simpool`): packing and launching.

Mirrors what DLDKD.get_sim_scores + compute_query2ctx_info + eval_epoch's fusion do in the reference
(method/model.py:307-329, method/eval.py:200-208,254) but keeps the gallery resident in a packed bf16
layout and never builds the (Nq, L, Nv) clip tensor.
"""
import os

import torch

from . import native

HIDDEN = 384
MAX_CLIPS = 128
# K1 with two videos per wave where they fit (dldkd_simpool_eval_pairs_bf16; profiles/r04/ablation_simpool_ragged.md)
PAIR_WAVES = os.environ.get("DLDKD_K1_PAIR_WAVES", "1") == "1"


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class PackedQueries:
    """bf16 queries in MFMA B-fragment order, one blob per branch; `bad` (Nq floats): 1 where a query vector of either branch
    has a NaN / Inf component (its scores become NaN in simpool_finish / rank_partials: it ranks last)."""

    def __init__(self, blobs, nq, bad=None):
        self.blobs = blobs
        self.nq = nq
        self.bad = bad


class PackedGallery:
    """Resident bf16 gallery: one blob per branch + lens + a visiting order (descending length)."""

    def __init__(self, blobs, lens, order, inv_order, nv, L):
        self.blobs, self.lens, self.order, self.inv_order, self.nv, self.L = blobs, lens, order, inv_order, nv, L

    @property
    def n_branches(self):
        return len(self.blobs)

    def pair_plan(self):
        """(pairs int32 GPU tensor (n_waves, 2), n_waves, n_paired) for dldkd_simpool_eval_pairs_bf16, built once per gallery (one
        copy of the lens to the host: part of packing, outside any timed region)."""
        if getattr(self, "_pairs", None) is None:
            host = pair_waves(self.lens[self.order.long()].cpu().numpy())
            self._pairs = (torch.from_numpy(host).to(self.lens.device), int(host.shape[0]), int((host[:, 1] >= 0).sum()))
        return self._pairs

    def scorer_waves(self):
        """Waves of one branch of the scorer launch (its grid is ceil(waves / 4) workgroups per branch and query range)."""
        if PAIR_WAVES and self.nv:
            n_waves, n_paired = self.pair_plan()[1:]
            if n_paired:
                return n_waves
        return self.nv


def pair_waves(sorted_lens, rows=MAX_CLIPS):
    """Two videos per scorer wave where they FILL it.  sorted_lens: clip counts in visiting order (descending, <= rows).  Video A of
    a pair ends on a 4-row boundary (the pooling routes whole 4-row lane groups) and a pair is only formed if it needs all eight
    16-row tiles (the pair loop is built for 8): round_up(len A, 4) + len B in (rows - 16, rows].  Greedy on the length histogram,
    longest first: a video takes the longest unplaced video that still fits behind it (best fit); everything else stays a wave of
    its own.  Videos of one length are interchangeable, so the plan is made per (len A, len B) class - a few hundred steps
    whatever the gallery size - and materialised with numpy.  Returns int32 (n_waves, 2) rows (posA, posB or -1) in wave order:
    long and full waves first, the grid's tail light."""
    import numpy as np
    sl = np.asarray(sorted_lens, dtype=np.int64)
    n = int(sl.shape[0])
    if n == 0:
        return np.zeros((0, 2), dtype=np.int32)
    if int(sl.max()) > rows or int(sl.min()) < 0 or (n > 1 and bool((sl[1:] > sl[:-1]).any())):
        raise native.NativeError("pair_waves: lengths must be descending and in [0, %d]" % rows)
    cnt = np.bincount(sl, minlength=rows + 1)
    start = n - np.cumsum(cnt)                      # positions of length a: start[a] .. start[a] + cnt[a] - 1
    front = [0] * (rows + 1)                        # taken from the front of a class (as A or alone) / from its back (as B)
    back = [0] * (rows + 1)
    cnt = cnt.tolist()
    pa, pb = [], []
    for a in range(rows, -1, -1):
        a4 = (a + 3) // 4 * 4
        while cnt[a] - front[a] - back[a] > 0:
            free_a = cnt[a] - front[a] - back[a]
            b, k = -1, 0
            if a > 0:
                for c in range(min(rows - a4, a), max(rows - 16 - a4, 0), -1):
                    free_c = cnt[c] - front[c] - back[c]
                    k = free_c // 2 if c == a else min(free_a, free_c)
                    if k > 0:
                        b = c
                        break
            first = int(start[a]) + front[a]
            if b < 0:
                pa.append(np.arange(first, first + free_a))
                pb.append(np.full(free_a, -1, dtype=np.int64))
                front[a] += free_a
            else:
                last = int(start[b]) + cnt[b] - 1 - back[b]
                pa.append(np.arange(first, first + k))
                pb.append(np.arange(last, last - k, -1))
                front[a] += k
                back[b] += k
    return np.stack([np.concatenate(pa), np.concatenate(pb)], axis=1).astype(np.int32)


def pack_queries(qs, normalize=True):
    """qs: list (one per branch) of (Nq, 384) tensors on the GPU."""
    L = native.lib()
    nq = qs[0].shape[0]
    blobs = []
    bad = torch.zeros((max(nq, 1) + 31) // 32 * 32, dtype=torch.float32, device=qs[0].device)
    for q in qs:
        if q.dim() != 2 or q.shape[1] != HIDDEN or q.shape[0] != nq:
            raise native.NativeError(f"queries must be (Nq, {HIDDEN}); got {tuple(q.shape)}")
        q = _f32c(q)
        blob = torch.empty(L.dldkd_packed_queries_bytes(nq), dtype=torch.uint8, device=q.device)
        native.check(L.dldkd_pack_queries_bf16(native.ptr(q), nq, int(normalize), native.ptr(blob), native.ptr(bad), native.stream()),
                     "pack_queries")
        blobs.append(blob)
    return PackedQueries(blobs, nq, bad)


def visit_order(lens):
    """(order, inv_order) int32: videos longest first, equal lengths in index order - the 4 waves of a scorer workgroup get
    similar lengths and the tail of the grid is light (one small counting-sort kernel)."""
    nv = lens.shape[0]
    order = torch.empty(nv, dtype=torch.int32, device=lens.device)
    inv = torch.empty(nv, dtype=torch.int32, device=lens.device)
    if nv:
        native.check(native.lib().dldkd_order_by_len_desc(native.ptr(lens.contiguous()), nv, native.ptr(order), native.ptr(inv),
                                                          native.stream()), "order_by_len_desc")
    return order, inv


def pack_gallery(gs, mask=None, normalize=True):
    """gs: list (one per branch) of (Nv, L, 384) GPU tensors; mask (Nv, L) 0/1 prefix mask or None."""
    L_ = native.lib()
    nv, L = gs[0].shape[0], gs[0].shape[1]
    if L > MAX_CLIPS:
        raise native.NativeError(f"at most {MAX_CLIPS} clips per video (config max_ctx_l); got {L}")
    dev = gs[0].device
    lens = torch.empty(max(nv, 1), dtype=torch.int32, device=dev)
    m = None if mask is None else _f32c(mask)
    blobs = []
    for g in gs:
        if g.dim() != 3 or g.shape[2] != HIDDEN or g.shape[0] != nv or g.shape[1] != L:
            raise native.NativeError(f"gallery must be (Nv, L, {HIDDEN}); got {tuple(g.shape)}")
        g = _f32c(g)
        blob = torch.empty(L_.dldkd_packed_gallery_bytes(nv, L), dtype=torch.uint8, device=dev)
        native.check(L_.dldkd_pack_gallery_bf16(native.ptr(g), native.ptr(m), nv, L, int(normalize), native.ptr(blob),
                                                native.ptr(lens), native.stream()), "pack_gallery")
        blobs.append(blob)
    lens = lens[:nv]
    order, inv = visit_order(lens)
    return PackedGallery(blobs, lens, order, inv, nv, L)


class GalleryPacker:
    """Streaming pack_gallery: the eval driver hands over one encoded batch at a time and the fp32 (Nv, L, 384)
    gallery of the reference (eval.py:139-175, 2 x 4.3 GB at TVR scale) never exists.  add() packs videos
    [v0, v0 + n) of every branch; finish() computes the visiting order and returns the PackedGallery."""

    def __init__(self, nv, L, n_branches, device, normalize=True, zero_fill=False, blobs=None):
        """zero_fill: the blobs start as zeros (a producer may then leave the rows past a video's last 16-row tile unwritten:
        zero_padded); blobs: buffers of an earlier packer of the SAME gallery (same videos, same lengths) that was zero-filled -
        their padding is still zero, whatever was written into the valid rows."""
        if L > MAX_CLIPS:
            raise native.NativeError(f"at most {MAX_CLIPS} clips per video (config max_ctx_l); got {L}")
        L_ = native.lib()
        self.nv, self.L, self.normalize, self.filled = nv, L, normalize, 0
        nbytes = L_.dldkd_packed_gallery_bytes(nv, L)
        if blobs is not None:
            want = torch.device(device)
            same = lambda b: b.device.type == want.type and (want.index is None or b.device.index == want.index)   # noqa: E731
            if len(blobs) != n_branches or any(b.numel() != nbytes or not same(b) for b in blobs):
                raise native.NativeError("GalleryPacker: the buffers handed in do not fit this gallery")
            self.blobs, self.zero_padded = list(blobs), True
        else:
            alloc = torch.zeros if zero_fill else torch.empty
            self.blobs = [alloc(nbytes, dtype=torch.uint8, device=device) for _ in range(n_branches)]
            self.zero_padded = bool(zero_fill)
        self.lens = torch.zeros(max(nv, 1), dtype=torch.int32, device=device)

    @property
    def Lp(self):
        return (self.L + 31) // 32 * 32

    def reserve(self, n, lc):
        """Claim videos [filled, filled + n) for a producer that writes the packed rows itself (the fused tower kernel);
        returns the first video index."""
        if self.filled + n > self.nv or lc > self.L:
            raise native.NativeError(f"GalleryPacker.reserve: batch of {n} x {lc} does not fit ({self.filled}/{self.nv} x {self.L})")
        v0 = self.filled
        self.filled += n
        return v0

    def add(self, gs, mask):
        n, lc = gs[0].shape[0], gs[0].shape[1]
        if len(gs) != len(self.blobs) or self.filled + n > self.nv or lc > self.L:
            raise native.NativeError(f"GalleryPacker.add: batch of {n} x {lc} does not fit ({self.filled}/{self.nv} x {self.L})")
        m = None if mask is None else _f32c(mask)
        for g, blob in zip(gs, self.blobs):
            if g.dim() != 3 or g.shape[2] != HIDDEN or g.shape[0] != n or g.shape[1] != lc:
                raise native.NativeError(f"gallery batch must be (n, L, {HIDDEN}); got {tuple(g.shape)}")
            native.check(native.lib().dldkd_pack_gallery_chunk_bf16(native.ptr(_f32c(g)), native.ptr(m), n, lc, int(self.normalize),
                                                                    native.ptr(blob), native.ptr(self.lens), self.filled, self.nv,
                                                                    self.L, native.stream()), "pack_gallery_chunk")
        self.filled += n

    def finish(self):
        if self.filled != self.nv:
            raise native.NativeError(f"GalleryPacker.finish: {self.filled} of {self.nv} videos packed")
        lens = self.lens[:self.nv]
        order, inv = visit_order(lens)
        return PackedGallery(self.blobs, lens, order, inv, self.nv, self.L)


def plan_query_split(nq, nv, n_branches, min_split=1):
    """(n_ranges, queries_per_range) the scorer would use / should be given for this problem (host-side cost model of
    dldkd_simpool_eval_plan: a 615-video shard is 308 workgroups on 256 CUs, so the queries are split to fill the chip)."""
    import ctypes
    n, per = ctypes.c_int(0), ctypes.c_int(0)
    native.check(native.lib().dldkd_simpool_eval_plan(nq, nv, n_branches, min_split, ctypes.byref(n), ctypes.byref(per)),
                 "simpool_eval_plan")
    return n.value, per.value


def simpool_partials(pq, pg, workspace=None, q_split=0, done=None):
    """Stage 1 (the dominant kernel): per-branch pooled scores into the workspace, transposed and in
    visiting order.  Returns the workspace tensor.  q_split = number of query ranges (0: chosen by the library);
    done = int32 GPU tensor of per-range arrival counters (zeroed by the caller) or None."""
    L_ = native.lib()
    nb = pg.n_branches
    if len(pq.blobs) != nb:
        raise native.NativeError("query / gallery branch count mismatch")
    need = L_.dldkd_simpool_eval_workspace_bytes(pq.nq, pg.nv, nb)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=pg.lens.device)
    if done is not None and (done.dtype != torch.int32 or done.numel() < max(q_split, 1)):
        raise native.NativeError("simpool_partials: `done` must hold one int32 per query range")
    if pq.nq and pg.nv and pg.scorer_waves() != pg.nv:
        pairs, n_waves, _ = pg.pair_plan()
        native.check(L_.dldkd_simpool_eval_pairs_bf16(native.ptr_array(pq.blobs), native.ptr_array(pg.blobs), native.ptr(pg.lens),
                                                      native.ptr(pg.order), native.ptr(pairs), n_waves, pq.nq, pg.nv, pg.L, nb,
                                                      int(q_split), native.ptr(done), native.ptr(workspace), native.stream()),
                     "simpool_eval_pairs")
    elif pq.nq and pg.nv:
        native.check(L_.dldkd_simpool_eval_bf16(native.ptr_array(pq.blobs), native.ptr_array(pg.blobs), native.ptr(pg.lens),
                                                native.ptr(pg.order), pq.nq, pg.nv, pg.L, nb, int(q_split), native.ptr(done),
                                                native.ptr(workspace), native.stream()), "simpool_eval")
    return workspace


def simpool_finish(workspace, pq, pg, w=(0.7, 0.3), want_fused=True, want_branches=False, q_range=None, out=None):
    """Stage 2: (Nq, Nv) outputs.  Returns (fused, s0, s1), each fp32 (Nq, Nv) or None.  q_range = (lo, hi) restricts
    the outputs to those query rows ((hi - lo, Nv) blocks); `out` = preallocated fused block."""
    L_ = native.lib()
    nb, nq, nv, dev = pg.n_branches, pq.nq, pg.nv, pg.lens.device
    lo, hi = (0, nq) if q_range is None else q_range
    n = hi - lo
    fused = (out if out is not None else torch.empty(n, nv, dtype=torch.float32, device=dev)) if want_fused else None
    s0 = torch.empty(n, nv, dtype=torch.float32, device=dev) if want_branches else None
    s1 = torch.empty(n, nv, dtype=torch.float32, device=dev) if (want_branches and nb == 2) else None
    if fused is not None and (tuple(fused.shape) != (n, nv) or fused.dtype != torch.float32):
        raise native.NativeError(f"simpool_finish: out must be fp32 ({n}, {nv})")
    if n and nv:
        native.check(L_.dldkd_simpool_finish_range(native.ptr(workspace), native.ptr(pg.inv_order), nq, nv, nb, float(w[0]),
                                                   float(w[1]), lo, hi, native.ptr(pq.bad), native.ptr(fused), native.ptr(s0),
                                                   native.ptr(s1), native.stream()), "simpool_finish")
    return fused, s0, s1


def rank_partials(workspace, pq, pg, gt_ptr, gt_idx, w=(0.7, 0.3)):
    """Ranks of the ground-truth videos straight from the scorer's partial planes (no (Nq, Nv) matrix is written or read).
    gt_ptr / gt_idx: the ground truth as CSR int32 GPU tensors (eval.gt_csr).  Returns int32 GPU tensor (3, 2, Nq):
    [kind: branch 0 / branch 1 / fused][rank of the best GT video (eval_q2m) / of the first listed one (t2v_map)]."""
    L_ = native.lib()
    nq, nv, dev = pq.nq, pg.nv, pg.lens.device
    counts = torch.empty(3, 2, nq, dtype=torch.int32, device=dev)
    if nq == 0:
        return counts
    if nv == 0:
        return counts.fill_(1)
    thr = torch.empty(6 * nq, dtype=torch.float32, device=dev)
    native.check(L_.dldkd_simpool_rank_partials(native.ptr(workspace), native.ptr(pg.inv_order), nq, nv, pg.n_branches, float(w[0]),
                                                float(w[1]), native.ptr(gt_ptr), native.ptr(gt_idx), native.ptr(pq.bad), native.ptr(thr),
                                                native.ptr(counts), native.stream()), "simpool_rank_partials")
    return torch.clamp_(counts.add_(1), max=nv + 1)


def shard_thresholds(workspace, pq, pg, gt_ptr, gt_idx, first_local, w=(0.7, 0.3)):
    """Local half 1 of sharded ranking from the partial planes: (thr, nan_flag) fp32 (3, 2, Nq) over THIS shard's ground-truth
    videos (gt_ptr / gt_idx: CSR of local GT indices; first_local int32 (Nq,): the first listed GT video is local).  -inf / 0
    where the shard holds none: all-reduce both with MAX."""
    L_ = native.lib()
    nq, dev = pq.nq, pg.lens.device
    thr = torch.full((3, 2, nq), float("-inf"), dtype=torch.float32, device=dev)
    flag = torch.zeros(3, 2, nq, dtype=torch.float32, device=dev)
    if nq and pg.nv:
        native.check(L_.dldkd_simpool_rank_partials_thr(native.ptr(workspace), native.ptr(pg.inv_order), nq, pg.nv, pg.n_branches,
                                                        float(w[0]), float(w[1]), native.ptr(gt_ptr), native.ptr(gt_idx),
                                                        native.ptr(first_local), native.ptr(pq.bad), native.ptr(thr), native.ptr(flag),
                                                        native.stream()), "simpool_rank_partials_thr")
    return thr, flag


def shard_counts(workspace, pq, pg, thr, w=(0.7, 0.3)):
    """Local half 2: counts int32 (3, 2, Nq) of this shard's videos scoring above the (all-reduced) thresholds, straight from
    the partial planes.  All-reduce with SUM."""
    L_ = native.lib()
    nq, dev = pq.nq, pg.lens.device
    counts = torch.zeros(3, 2, nq, dtype=torch.int32, device=dev)
    if nq:
        native.check(L_.dldkd_simpool_rank_partials_count(native.ptr(workspace) if pg.nv else None, nq, pg.nv, pg.n_branches, float(w[0]),
                                                          float(w[1]), native.ptr(thr.contiguous()), native.ptr(counts), native.stream()),
                     "simpool_rank_partials_count")
    return counts


def simpool_eval(pq, pg, w=(0.7, 0.3), want_fused=True, want_branches=False, workspace=None):
    """Pooled cosine/dot scores of every query against every video.

    Returns (fused, s0, s1): (Nq, Nv) fp32 tensors or None.  fused = w[0]*s0 + w[1]*s1 (eval.py:254),
    or s0 alone for a single-branch model.
    """
    ws = simpool_partials(pq, pg, workspace)
    return simpool_finish(ws, pq, pg, w, want_fused, want_branches)
