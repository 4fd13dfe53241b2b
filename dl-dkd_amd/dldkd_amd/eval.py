"""Eval driver: encode the gallery, score every query against every video, rank, R@K.

Same function names, arguments and return values as reference method/eval.py:43-263 for the part that
is on the hot path.  Differences in HOW (not in what is returned):
  * the gallery is packed once into the resident bf16 layout (the reference re-normalises it for every
    50-query chunk, model.py:319) and all queries are scored in ONE launch of the MFMA scorer;
  * ranks are counted on the GPU (rank = 1 + #scores above the best ground-truth video) instead of
    np.argsort per query + a pure-Python AP loop (eval.py:69-83,97-111).
"""
import contextlib
import os
import logging
import weakref

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import native, ops, scoring
from .data import collate_frame_val, collate_text_val, host_threads

logger = logging.getLogger(__name__)


def get_gt(video_metas, query_metas):
    """cap_id 'vid#...' is a caption of video 'vid' (eval.py:43-57).  Returns (v2t_gt, t2v_gt)."""
    pos = {}
    for i, v in enumerate(video_metas):
        pos.setdefault(v, []).append(i)
    v2t_gt = [[] for _ in video_metas]
    for qi, cap in enumerate(query_metas):
        for vi in pos.get(cap.split("#", 1)[0], ()):
            v2t_gt[vi].append(qi)
    t2v_gt = {}
    for vi, qs in enumerate(v2t_gt):
        for qi in qs:
            t2v_gt.setdefault(qi, []).append(vi)
    return v2t_gt, t2v_gt


def _gt_csr(t2v_gt, n_q, device):
    ptr = np.zeros(n_q + 1, np.int32)
    idx = []
    for q in range(n_q):
        g = t2v_gt.get(q, []) if isinstance(t2v_gt, dict) else t2v_gt[q]
        idx.extend(g)
        ptr[q + 1] = len(idx)
    idx = np.asarray(idx if idx else [0], np.int32)
    return torch.from_numpy(ptr).to(device), torch.from_numpy(idx).to(device)


def gt_csr(t2v_gt, n_q, device):
    """Ground truth as CSR (ptr, idx) on the device: build it once when several score matrices are ranked against the
    same ground truth (eval_epoch ranks three) and pass it to gt_ranks_gpu / cal_perf as `csr`."""
    return _gt_csr(t2v_gt, n_q, device)


def gt_ranks_gpu(scores, t2v_gt, csr=None):
    """scores: (Nq, Nv) fp32 GPU similarity (higher = better).  Returns (rank_best, rank_first) int32 GPU."""
    L = native.lib()
    scores = scores.contiguous()
    nq, nv = scores.shape
    ptr, idx = csr if csr is not None else _gt_csr(t2v_gt, nq, scores.device)
    rb = torch.empty(nq, dtype=torch.int32, device=scores.device)
    rf = torch.empty(nq, dtype=torch.int32, device=scores.device)
    native.check(L.dldkd_rank_gt(native.ptr(scores), nq, nv, native.ptr(ptr), native.ptr(idx), native.ptr(rb),
                                 native.ptr(rf), native.stream()), "rank_gt")
    return rb, rf


def _recalls(ranks, n_q):
    r = ranks.astype(np.int64)
    out = [100.0 * int((r <= k).sum()) / n_q for k in (1, 5, 10, 100)]
    return (out[0], out[1], out[2], out[3], float(np.median(r)), float(r.mean()))


def eval_q2m(scores, q2m_gts):
    """(r1, r5, r10, r100, medr, meanr) as eval.py:59-94.  `scores` is an ERROR matrix (lower = better,
    the reference passes -1 * similarity); numpy or torch, any device - it is ranked on the GPU."""
    t = torch.as_tensor(scores)
    if not t.is_cuda:
        t = t.cuda()
    rb, _ = gt_ranks_gpu(-t.float(), q2m_gts)
    return _recalls(rb.cpu().numpy(), t.shape[0])


def t2v_map(c2i, t2v_gts):
    """mean AP over queries using the first GT video only (eval.py:97-111)."""
    t = torch.as_tensor(c2i)
    if not t.is_cuda:
        t = t.cuda()
    _, rf = gt_ranks_gpu(-t.float(), t2v_gts)
    return float((1.0 / rf.cpu().numpy().astype(np.float64)).mean())


def cal_perf(t2v_all_errors, t2v_gt, test=False, csr=None):
    """eval.py:223-234: logs and returns (r1, r5, r10, r100, medr, meanr, mAP)."""
    t = torch.as_tensor(t2v_all_errors)
    if not t.is_cuda:
        t = t.cuda()
    rb, rf = gt_ranks_gpu(-t.float(), t2v_gt, csr)
    r1, r5, r10, r100, medr, meanr = _recalls(rb.cpu().numpy(), t.shape[0])
    m = float((1.0 / rf.cpu().numpy().astype(np.float64)).mean())
    logging.info(" * Text to Video:")
    logging.info(" * r_1_5_10_100: {}".format([round(r1, 1), round(r5, 1), round(r10, 1), round(r100, 1)]))
    logging.info(" * recall sum: {}".format(round(r1 + r5 + r10 + r100, 1)))
    logging.info(" * mAP: {}".format(round(m, 4)))
    logging.info(" * " + "-" * 10)
    return (r1, r5, r10, r100, medr, meanr, m)


def perf_from_ranks(rank_best, rank_first, n_q):
    """cal_perf's seven numbers (and its log lines) from ranks that are already known."""
    r1, r5, r10, r100, medr, meanr = _recalls(rank_best, n_q)
    m = float((1.0 / rank_first.astype(np.float64)).mean())
    logging.info(" * Text to Video:")
    logging.info(" * r_1_5_10_100: {}".format([round(r1, 1), round(r5, 1), round(r10, 1), round(r100, 1)]))
    logging.info(" * recall sum: {}".format(round(r1 + r5 + r10 + r100, 1)))
    logging.info(" * mAP: {}".format(round(m, 4)))
    logging.info(" * " + "-" * 10)
    return (r1, r5, r10, r100, medr, meanr, m)


# ---- device-resident raw features of the evaluation datasets.  train() evaluates the SAME validation videos and captions after
# every epoch; their raw features never change, only the towers do.  From host memory an eval epoch is PCIe-bound 10-35 : 1
# (0.6-2.1 s of pinned H2D for the TVR gallery against 0.044 s of GPU work), so the loader batches of the first
# pass stay on the device (34 GB of fp32 clips for TVR's 21,793 videos: 12 % of the 288 GB) and later passes replay them without
# touching the DataLoader.  opt.eval_feature_cache = False turns it off; opt.eval_feature_cache_gb caps it (default 96).
_FEATURE_CACHE = weakref.WeakKeyDictionary()          # dataset -> {kind: [device batches]}


def clear_feature_cache():
    _FEATURE_CACHE.clear()
    _GT_CACHE.clear()


def _cache_bytes():
    return sum(k["bytes"] if isinstance(k, dict) else k.table.nbytes() + sum(b.numel() for b in (getattr(k, "gallery_blobs", None) or []))
               for d in _FEATURE_CACHE.values() for k in d.values())


def _cached_batches(dataset, kind, n_items, opt, make_loader, to_device):
    """Generator of the dataset's loader batches as device tuples: from the cache when this dataset (`kind` names the part of
    it: "text", "video", or a rank's ("video", lo, hi) shard) has been through before, otherwise from the DataLoader (and into the
    cache while it fits the cap)."""
    dev = torch.device(opt.device)
    use = bool(getattr(opt, "eval_feature_cache", True)) and dev.type == "cuda"
    try:
        slot = _FEATURE_CACHE.setdefault(dataset, {}) if use else None
    except TypeError:                                   # a dataset object that cannot be weakly referenced
        slot = None
    if slot is not None and kind in slot and slot[kind]["complete"] and slot[kind]["n"] == n_items \
            and slot[kind]["device"] == dev:
        # a DataLoader iterator draws its base seed from the global CPU generator when it is created: keep that draw, so that
        # a training run sees the same random stream (triplet negatives, shuffles) with and without the cache
        torch.empty((), dtype=torch.int64).random_()
        yield from slot[kind]["batches"]
        return
    cap = float(getattr(opt, "eval_feature_cache_gb", 96.0)) * 1e9
    entry = {"batches": [], "bytes": 0, "complete": False, "n": n_items, "device": dev}
    keep = slot is not None
    for batch in make_loader():
        b = to_device(batch)
        if keep:
            nbytes = sum(t.numel() * t.element_size() for t in b if torch.is_tensor(t))
            total = _cache_bytes() + entry["bytes"] + nbytes
            if total > cap:
                keep, entry = False, {"batches": [], "bytes": 0, "complete": False, "n": n_items, "device": dev}
            else:
                entry["batches"].append(b)
                entry["bytes"] += nbytes
        yield b
    if keep:
        entry["complete"] = True
        slot[kind] = entry


# ---- the gallery's raw features in their RESIDENT form (throughput mode).  What the input projection consumes is bf16 rows and
# the rows' LayerNorm statistics (K4 rounds the fp32 features to bf16 on their way to the MFMA and sums the statistics on the
# side); so the first pass converts every loader batch ONCE into a ragged table of exactly that (ops.ResidentRows: no padding
# rows, 6 KB per clip instead of 12: 10 GB for TVR's 1.66 M clips against 34 GB of padded fp32 batches), and every gallery
# encode - the first one included - is the input projection over the whole table (K4b: dense 128-row tiles, no conversions in
# the k-loop, no per-batch tail) + the fused tower kernel over the whole gallery: two launches per chunk of RESIDENT_CHUNK_ROWS
# clips, no DataLoader, no H2D, no host-side padding or concatenation of batches, tables planned once.
RESIDENT_FEATURES = True          # throughput-mode eval_epoch keeps the gallery's raw features as ops.ResidentRows
RESIDENT_RAGGED_INGEST = True       # first pass: items -> pinned staging ring -> table rows (no padded host batches)
RESIDENT_STREAM_ROWS = 1 << 17      # clips staged per encode when the table is not kept (256 CUs x 4 tiles of 128 rows)
RESIDENT_CHUNK_ROWS = 1 << 21       # clips per chunk: the projection's fp32 output of a chunk is 2 x 3.2 GB


class ResidentGallery:
    def __init__(self, K, device):
        self.table = ops.ResidentRows(K, device)
        self.metas, self.complete, self.chunks = [], False, None

    def plan(self, device):
        """Per chunk of whole videos: (first video, #videos, first row, end row, lens int32 GPU, row0 int32 GPU relative to the
        chunk, slot table of the tower kernel) - planned and uploaded once."""
        lens = np.asarray(self.table.lens, dtype=np.int64)
        start = np.concatenate([[0], np.cumsum(lens)])
        self.lens_host, self.chunks, va = lens, [], 0
        while va < len(lens):
            vb = int(np.searchsorted(start, start[va] + RESIDENT_CHUNK_ROWS, side="right")) - 1
            vb = min(max(vb, va + 1), len(lens))
            l = lens[va:vb]
            meta = np.concatenate([l, start[va:vb] - start[va]]).astype(np.int32)
            meta_d = torch.from_numpy(meta).to(device)
            items = torch.from_numpy(ops.plan_tower_items(l)).to(device)
            self.chunks.append((va, vb - va, int(start[va]), int(start[vb]), meta_d[:vb - va], meta_d[vb - va:], items))
            va = vb
        self.lens_dev = torch.from_numpy(lens.astype(np.int32)).to(device)


def _items_as_they_are(items):
    return items


def _collate_ragged_rows(items):
    """[(feat (len, D), idx, id)] -> (rows (sum len, D) fp32, lens, ids): the loader batch a WORKER process hands over as ONE tensor (an
    identity collate would pass every item's tensor through shared memory: hundreds of descriptors per batch in flight)."""
    feats = [torch.as_tensor(it[0], dtype=torch.float32) for it in items]
    return (torch.cat(feats, 0) if feats else torch.zeros(0, 0)), [int(f.shape[0]) for f in feats], [it[2] for it in items]


def _resident_context_info(model, eval_dataset, opt, loader, owner, kind):
    """compute_context_info(keep_frame_feats=False) through a resident feature table; None when the path does not apply (the
    caller then encodes padded fp32 super-batches as before).  With the feature cache on, the table of the first pass stays (up to
    opt.eval_feature_cache_gb) and later passes replay it; without it (or past the cap) the table is a staging buffer that is
    encoded and emptied every RESIDENT_STREAM_ROWS clips."""
    from .model import _cfg_get
    dev = torch.device(opt.device)
    if not (dev.type == "cuda" and model.resident_encode_ok()):
        return None
    n = len(eval_dataset)
    L = int(_cfg_get(model.config, "max_ctx_l"))
    if n == 0 or L > 128:
        return None
    slot = None
    if bool(getattr(opt, "eval_feature_cache", True)):
        try:
            slot = _FEATURE_CACHE.setdefault(owner, {})
        except TypeError:                                   # a dataset object that cannot be weakly referenced
            slot = None
    key = ("resident", kind)
    lmax_host = None
    res = slot.get(key) if slot is not None else None
    # opt.eval_resident_shard: the table persisted as ONE file (ingest.save_resident / load_resident: fp16 rows + statistics + ids,
    # read back through a pinned ring at PCIe rate) - a fresh process starts from it instead of re-reading the fp32 features item
    # by item through the loader (utils/basic_utils.py:9-68); written behind the first pass that built the whole table.
    shard = getattr(opt, "eval_resident_shard", None) if kind == "video" else None
    if res is None and shard and os.path.exists(shard):
        from . import ingest
        table, ids = ingest.load_resident(shard, dev)
        if table.K != int(model.visual_input_proj.net[1].weight.shape[1]) or len(ids) != n or max(table.lens, default=0) > L:
            raise native.NativeError(f"eval_resident_shard {shard}: {len(ids)} items of width {table.K} do not match the gallery "
                                     f"({n} videos, width {int(model.visual_input_proj.net[1].weight.shape[1])}, max_ctx_l {L})")
        res = ResidentGallery(table.K, dev)
        res.table, res.metas, res.complete = table, ids, True
        res.plan(dev)
        if slot is not None:
            slot[key] = res
    if res is not None and res.complete and len(res.metas) == n and res.table.device == dev:
        # the packed gallery's buffers live with the resident table: zero-filled once, re-encoded in place every epoch (a dataset's
        # lengths do not change, so the padding the scorers never load is never written again: 4.29 -> 2.56 GB per TVR gallery).
        # A context dict of an earlier epoch of the SAME dataset therefore shows the newest encode.
        blobs = getattr(res, "gallery_blobs", None)
        try:
            packer = scoring.GalleryPacker(n, L, 2, dev, blobs=blobs) if blobs is not None else None
        except native.NativeError:
            packer = None
        if packer is None:
            packer = scoring.GalleryPacker(n, L, 2, dev, zero_fill=True)
            res.gallery_blobs = packer.blobs
        torch.empty((), dtype=torch.int64).random_()        # the DataLoader iterator's base-seed draw (see _cached_batches)
        with torch.no_grad():
            model.encode_resident_into(packer, res)
        metas, lens_all = list(res.metas), res.lens_dev
        lmax_host = int(res.lens_host.max(initial=0))       # (known on the host: no read-back in front of the queries' encode)
    else:
        packer = scoring.GalleryPacker(n, L, 2, dev)
        cap = float(getattr(opt, "eval_feature_cache_gb", 96.0)) * 1e9
        other = _cache_bytes()
        keep = slot is not None
        res = ResidentGallery(int(model.visual_input_proj.net[1].weight.shape[1]), dev)
        metas, lens_parts = [], []

        def encode_table():
            res.plan(dev)
            with torch.no_grad():
                model.encode_resident_into(packer, res)
            lens_parts.append(res.lens_dev)

        with host_threads():
            if RESIDENT_RAGGED_INGEST:
                # The items as the dataset hands them out (an identity collate in the same DataLoader: same order, same workers,
                # same single draw from the global generator), each clip row copied ONCE into a pinned staging ring and uploaded
                # from there into the table (ResidentRows.append_rows) - no padded host batch is ever built: building and
                # freeing 315-MB pageable batches (pad_sequence, page faults, munmap) was 2/3 of the first pass from host memory
                from torch.utils.data import DataLoader
                from .data import _PinnedAppender
                chunks, pending = [], []
                fill = [0, 0]                               # rows handed to the table / rows of the items registered with it

                def drain(final=False):
                    """uploaded row chunks -> the table; the lengths of the items a chunk completes go with it"""
                    for c in chunks:
                        fill[0] += int(c.shape[0])
                        k, tot = 0, fill[1]
                        while k < len(pending) and tot + pending[k] <= fill[0]:
                            tot += pending[k]
                            k += 1
                        res.table.append_rows(c, pending[:k])
                        fill[1] = tot
                        del pending[:k]
                    chunks.clear()
                    if final and pending:
                        raise native.NativeError("eval: resident ingest lost rows")
                app = _PinnedAppender(dev, chunks)
                workers = int(loader.num_workers)
                raw = DataLoader(loader.dataset, batch_size=loader.batch_size, shuffle=False, num_workers=workers,
                                 collate_fn=_collate_ragged_rows if workers else _items_as_they_are)
                for items in raw:
                    if workers:                             # the batch's rows as one tensor, concatenated by the worker
                        rows, lens_b, ids_b = items
                        if lens_b and (rows.dim() != 2 or rows.shape[1] != res.table.K or max(lens_b) > L):
                            raise native.NativeError("eval: a gallery item does not match the model's feature width / max_ctx_l")
                        if lens_b:
                            app.add(rows)
                        pending.extend(lens_b)
                        metas.extend(ids_b)
                    else:
                        for feat, _idx, vid in items:
                            feat = torch.as_tensor(feat)
                            if feat.dim() != 2 or feat.shape[1] != res.table.K or feat.shape[0] > L:
                                raise native.NativeError("eval: a gallery item does not match the model's feature width / max_ctx_l")
                            app.add(feat)
                            pending.append(int(feat.shape[0]))
                            metas.append(vid)
                    drain()
                    if keep and other + res.table.nbytes() + app.rows * res.table.K * 4 > cap:
                        keep = False
                    if not keep and res.table.rows + app.rows >= RESIDENT_STREAM_ROWS:
                        app.flush()                        # whole items only in the table before it is encoded and emptied
                        drain()
                        encode_table()
                        res.table.clear()
                        fill[0] = fill[1] = 0
                app.flush(final=True)
                drain(final=True)
            else:
                for batch in loader:
                    feat = batch[0].to(dev, non_blocking=True)
                    lens_h = (batch[1] > 0).sum(1).numpy()
                    if feat.shape[-1] != res.table.K or int(lens_h.max(initial=0)) > L:
                        raise native.NativeError("eval: a gallery batch does not match the model's feature width / max_ctx_l")
                    res.table.append(feat.float(), lens_h)
                    metas.extend(batch[-1])
                    if keep and other + res.table.nbytes() > cap:
                        keep = False
                    if not keep and res.table.rows >= RESIDENT_STREAM_ROWS:
                        encode_table()
                        res.table.clear()
        if keep:
            res.metas, res.complete = metas, True
            slot[key] = res
            if shard and not os.path.exists(shard):
                from . import ingest
                torch.cuda.synchronize(dev)
                ingest.save_resident(shard, res.table, metas)
                logger.info(f"resident gallery features saved to {shard} ({res.table.nbytes() / 1e9:.2f} GB)")
        if res.table.rows or not lens_parts:
            encode_table()
        lens_all = lens_parts[0] if len(lens_parts) == 1 else torch.cat(lens_parts)
    if lmax_host is None:
        lmax_host = int(lens_all.max().item()) if lens_all.numel() else 0
    vmask = (torch.arange(lmax_host, device=dev).unsqueeze(0) < lens_all.unsqueeze(1)).float()
    return dict(video_metas=metas, inher_frame_feat=None, explore_frame_feat=None, teacher_frame_feat=None,
                video_mask=vmask, _packed=packer.finish())


CONTEXT_SUPER_BATCH = 1024
TOWER_ITEM_BUDGET = 768        # slot groups of the fused tower kernel per super-batch: 3 rounds of the 256 CUs per branch


def _tiles(lens_list):
    return int(sum(int(((l + 31) // 32).sum()) for l in lens_list))


def _take_for_budget(lens, budget):
    """The longest prefix of the pending videos whose slot groups (ops.plan_tower_items) fit the budget."""
    cum = np.cumsum((lens + 31) // 32)
    n = int(np.searchsorted(cum, 4 * budget, side="right"))
    while n > 1:
        over = len(ops.plan_tower_items(lens[:n])) - budget
        if over <= 0:
            break
        n -= max(1, over)                  # a video is at least one tile: dropping `over` videos frees at most 4 x over slots
    return max(n, 1)


def compute_context_info(model, eval_dataset, opt, keep_frame_feats=True, cache_owner=None, cache_kind="video"):
    """Encode the gallery in batches of eval_context_bsz, zero-pad to the global max length, concatenate
    (eval.py:114-175).  Adds `_packed`: the resident bf16 gallery the scorer consumes.

    keep_frame_feats=False (what eval_epoch uses): every encoded batch is packed straight into the resident bf16
    gallery and dropped - the fp32 (Nv, Lmax, 384) tensors of the reference's dict (2 x 4.3 GB at TVR scale) are
    never formed and `inher_frame_feat` / `explore_frame_feat` are None."""
    from .model import _cfg_get
    model.eval()
    loader = DataLoader(eval_dataset, collate_fn=collate_frame_val, batch_size=opt.eval_context_bsz,
                        num_workers=opt.num_workers, shuffle=False, pin_memory=opt.pin_memory)
    metas, inh, exp, masks = [], [], [], []
    packer = None
    pend_f, pend_m, pend_l, pend_n = [], [], [], 0
    fused_path = not keep_frame_feats and getattr(model, "fast_input_proj", False) and ops.TOWER_SEQ
    if fused_path and RESIDENT_FEATURES:
        info = _resident_context_info(model, eval_dataset, opt, loader, cache_owner if cache_owner is not None else eval_dataset,
                                      cache_kind)
        if info is not None:
            return info

    def split_pending(take):
        """(first `take` pending videos, the rest), cutting a loader batch in two where the boundary falls inside it"""
        head, tail, n = ([], [], []), ([], [], []), 0
        for f, m_, l in zip(pend_f, pend_m, pend_l):
            k = min(max(take - n, 0), f.shape[0])
            if k > 0:
                head[0].append(f[:k]); head[1].append(m_[:k]); head[2].append(l[:k])
            if k < f.shape[0]:
                tail[0].append(f[k:]); tail[1].append(m_[k:]); tail[2].append(l[k:])
            n += f.shape[0]
        return head, tail

    def flush(take=None):
        """encode the pending loader batches (their first `take` videos) as ONE super-batch (zero-padded to its longest video;
        padded clips are masked out of attention exactly): at eval_context_bsz = 200 a batch is 200 workgroups of 128 rows on 256
        CUs, so every tower kernel runs a single partly-filled round; 1024 videos give four full ones"""
        nonlocal pend_f, pend_m, pend_l, pend_n, packer
        if not pend_f:
            return
        rest = None
        if take is not None and take < pend_n:
            (pend_f, pend_m, pend_l), rest = split_pending(take)
            pend_n = take
        try:
            flush_all()
        finally:
            if rest is not None:
                pend_f, pend_m, pend_l = rest
                pend_n = sum(f.shape[0] for f in pend_f)

    def flush_all():
        nonlocal pend_f, pend_m, pend_l, pend_n, packer
        if len(pend_f) == 1:
            feat, mask = pend_f[0], pend_m[0]
        else:
            lmax = max(f.shape[1] for f in pend_f)
            if getattr(model, "fast_input_proj", False):         # whole 32-row groups: the input projection then skips the padding
                lmax = min(-(-lmax // 32) * 32, max(int(_cfg_get(model.config, "max_ctx_l")), lmax))
            feat = pend_f[0].new_zeros(pend_n, lmax, pend_f[0].shape[2])
            mask = pend_m[0].new_zeros(pend_n, lmax)
            o = 0
            for f, m_ in zip(pend_f, pend_m):
                feat[o:o + f.shape[0], :f.shape[1]] = f
                mask[o:o + f.shape[0], :f.shape[1]] = m_
                o += f.shape[0]
        if not keep_frame_feats and getattr(model, "fast_input_proj", False):
            # throughput mode: input projection + ONE fused tower kernel that writes the packed bf16 gallery rows
            if packer is None:
                packer = scoring.GalleryPacker(len(eval_dataset), int(_cfg_get(model.config, "max_ctx_l")),
                                               2 if model.double_branch else 1, feat.device)
            if model.encode_context_into(packer, feat, mask, lens_host=np.concatenate(pend_l)):
                masks.append(mask)
                pend_f, pend_m, pend_l, pend_n = [], [], [], 0
                return
        gi, ge = model.encode_context(feat, mask)
        if keep_frame_feats:
            # the reference's dict holds zeros beyond each LOADER batch's own longest video (cat_tensor, eval.py:139-155);
            # positions between a video's length and that maximum keep the towers' output for padded clips, as there
            o = 0
            for f in pend_f:
                if f.shape[1] < gi.shape[1]:
                    gi[o:o + f.shape[0], f.shape[1]:] = 0
                    if ge is not None:
                        ge[o:o + f.shape[0], f.shape[1]:] = 0
                o += f.shape[0]
            inh.append(gi)
            exp.append(ge)
        else:
            if packer is None:
                packer = scoring.GalleryPacker(len(eval_dataset), int(_cfg_get(model.config, "max_ctx_l")),
                                               2 if model.double_branch else 1, gi.device)
            packer.add([gi, ge] if model.double_branch else [gi], mask)
        masks.append(mask)
        pend_f, pend_m, pend_l, pend_n = [], [], [], 0

    def to_device(batch):       # (features, mask, lengths on the host - the loader's mask is a CPU tensor -, metas)
        return (batch[0].to(opt.device, non_blocking=True), batch[1].to(opt.device, non_blocking=True),
                (batch[1] > 0).sum(1).numpy(), list(batch[-1]))

    with torch.no_grad(), host_threads():
        for feat_d, mask_d, lens_h, meta in _cached_batches(cache_owner if cache_owner is not None else eval_dataset, cache_kind,
                                                             len(eval_dataset), opt, lambda: loader, to_device):
            metas.extend(meta)
            pend_l.append(lens_h)
            pend_f.append(feat_d)
            pend_m.append(mask_d)
            pend_n += feat_d.shape[0]
            if fused_path:
                # the fused tower kernel runs one workgroup per (four 32-clip slots, branch) and owns a CU: cut the super-batches
                # where the planned workgroups fill whole rounds of the chip (1024 ragged videos planned to 773 slot groups = 6.04
                # rounds per branch pair: a seventh round for 10 workgroups, +13 % of the kernel)
                while pend_n and _tiles(pend_l) >= 4 * TOWER_ITEM_BUDGET:
                    flush(_take_for_budget(np.concatenate(pend_l), TOWER_ITEM_BUDGET))
            elif pend_n >= CONTEXT_SUPER_BATCH:
                flush()
        flush()

    def cat(tensors):
        lmax = max(t.shape[1] for t in tensors)
        out = tensors[0].new_zeros((sum(t.shape[0] for t in tensors), lmax) + tuple(tensors[0].shape[2:]))
        o = 0
        for t in tensors:
            out[o:o + t.shape[0], :t.shape[1]] = t
            o += t.shape[0]
        return out

    if not keep_frame_feats:
        if packer is None:          # empty gallery (a rank whose shard is empty: more ranks than videos)
            packer = scoring.GalleryPacker(0, int(_cfg_get(model.config, "max_ctx_l")), 2 if model.double_branch else 1,
                                           torch.device(opt.device))
        vmask = cat(masks) if masks else torch.zeros(0, 0, device=torch.device(opt.device))
        return dict(video_metas=metas, inher_frame_feat=None, explore_frame_feat=None, teacher_frame_feat=None,
                    video_mask=vmask, _packed=packer.finish())
    info = dict(video_metas=metas, inher_frame_feat=cat(inh),
                explore_frame_feat=cat(exp) if model.double_branch else None,
                teacher_frame_feat=None, video_mask=cat(masks))
    gs = [info["inher_frame_feat"]] + ([info["explore_frame_feat"]] if model.double_branch else [])
    info["_packed"] = scoring.pack_gallery(gs, info["video_mask"])
    return info


QUERY_SUPER_BATCH = 2048


def _encode_all_queries(model, eval_dataset, opt):
    """Encode every query; rows come out in query_metas order (per-loader-batch length-sorted, data_provider.py:153).
    Loader batches (eval_query_bsz = 50 in the reference's scripts) are grouped into super-batches of up to
    QUERY_SUPER_BATCH rows, zero-padded to a common word count, and encoded in ONE pass: padded words are masked out
    of attention and pooling exactly (exp(-10000) and exp(-1e10) are 0 in fp32), so the vectors are those of the
    per-batch calls while the launch count drops ~40x."""
    loader = DataLoader(eval_dataset, collate_fn=collate_text_val, batch_size=opt.eval_query_bsz,
                        num_workers=opt.num_workers, shuffle=False, pin_memory=opt.pin_memory)
    metas, qi, qe = [], [], []
    pend_f, pend_m, pend_n = [], [], 0

    def flush():
        nonlocal pend_f, pend_m, pend_n
        if not pend_f:
            return
        lw = max(f.shape[1] for f in pend_f)
        feat = pend_f[0].new_zeros(pend_n, lw, pend_f[0].shape[2])
        mask = pend_m[0].new_zeros(pend_n, lw)
        o = 0
        for f, m_ in zip(pend_f, pend_m):
            feat[o:o + f.shape[0], :f.shape[1]] = f
            mask[o:o + f.shape[0], :f.shape[1]] = m_
            o += f.shape[0]
        a, b = model.encode_query(feat, mask)
        qi.append(a.reshape(pend_n, -1))
        if b is not None:
            qe.append(b.reshape(pend_n, -1))
        pend_f, pend_m, pend_n = [], [], 0

    def to_device(batch):
        return (batch[0].to(opt.device, non_blocking=True), batch[1].to(opt.device, non_blocking=True), list(batch[-1]))

    with torch.no_grad(), host_threads():
        for feat_d, mask_d, meta in _cached_batches(eval_dataset, "text", len(eval_dataset), opt, lambda: loader, to_device):
            metas.extend(meta)
            pend_f.append(feat_d)
            pend_m.append(mask_d)
            pend_n += feat_d.shape[0]
            if pend_n >= QUERY_SUPER_BATCH:
                flush()
        flush()
    qs = [torch.cat(qi, 0)] + ([torch.cat(qe, 0)] if model.double_branch else [])
    return metas, qs


def score_queries(model, eval_dataset, opt, ctx_info):
    """GPU-resident form of compute_query2ctx_info: (fused, inher, explore, query_metas), tensors on the GPU."""
    model.eval()
    metas, qs = _encode_all_queries(model, eval_dataset, opt)
    fused, s0, s1 = model.pooled_scores(qs, ctx_info["_packed"], want_branches=True)
    return fused, s0, s1, metas


def compute_query2ctx_info(model, eval_dataset, opt, ctx_info):
    """(inher_scores, explore_scores, None, query_metas) with numpy (Nq, Nv) fp32 matrices, rows in
    query_metas order = per-batch length-sorted order (eval.py:177-219)."""
    _, s0, s1, metas = score_queries(model, eval_dataset, opt, ctx_info)
    return s0.cpu().numpy().copy(), (s1.cpu().numpy().copy() if s1 is not None else None), None, metas


def _dist_world():
    from . import comm
    return comm.info()


def gallery_ids(dataset):
    """Video ids of a gallery dataset WITHOUT reading features: the `video_ids` attribute of the reference's
    VisDataSet4DLDKD (data_provider.py:270-275), else `ids`, else a `get_video_id(i)` accessor.  Last resort (with a
    warning): item [2] of every sample, which loads every video's features - on every rank of a sharded eval."""
    for attr in ("video_ids", "ids"):
        ids = getattr(dataset, attr, None)
        if ids is not None:
            return list(ids)
    fn = getattr(dataset, "get_video_id", None)
    if fn is not None:
        return [fn(i) for i in range(len(dataset))]
    import warnings
    warnings.warn("gallery dataset exposes neither video_ids / ids nor get_video_id(i): reading every item to collect the ids "
                  "(loads all features on this rank)", stacklevel=2)
    return [dataset[i][2] for i in range(len(dataset))]


def eval_epoch_sharded(model, val_video_dataset, val_text_dataset, opt, test=False):
    """eval_epoch with the gallery sharded by video over the ranks of the current communicator (comm.current(); config C4).

    Rank r encodes and keeps videos [r*S, (r+1)*S) only (its features are the only ones it reads), every rank encodes all
    queries, scores them against its shard (the scorer's two partial planes, nothing else), and the ranks of all three score
    kinds come from dist.sharded_ranks_from_partials: thresholds over the local ground-truth videos straight from the planes,
    all-reduce(MAX), counts of local videos above them straight from the planes, all-reduce(SUM).  No (Nq, Nv / N) score matrix
    is written on any rank and nothing but 2 x (6 Nq) numbers crosses xGMI; R@K is exactly that of the unsharded evaluation (a
    caption whose video is not in the gallery ranks n_videos + 1, as in the unsharded path).  Returns SumR of the fused scores
    on every rank."""
    from torch.utils.data import Subset
    from . import dist as ddist
    from . import comm as _comm
    rank, world = _dist_world()
    use_collectives = _comm.current() is not None
    model.eval()
    n_videos = len(val_video_dataset)
    lo, hi, _ = ddist.shard_range(n_videos, rank, world)
    from . import ops
    guard = eval_precision_mode(opt, test) == "throughput" and torch.device(getattr(opt, "device", "cuda")).type == "cuda"
    if guard:
        ops.nonfinite_flag(opt.device).zero_()
    with eval_precision(model, opt, test):
        # (the shard is a fresh Subset object every epoch: its cached features are filed under the gallery dataset itself)
        ctx = compute_context_info(model, Subset(val_video_dataset, range(lo, hi)), opt, keep_frame_feats=False,
                                   cache_owner=val_video_dataset, cache_kind=("video", lo, hi))
        query_metas, qs = _encode_all_queries(model, val_text_dataset, opt)
    video_metas = gallery_ids(val_video_dataset) if world > 1 else ctx["video_metas"]
    _, t2v_gt = get_gt(video_metas, query_metas)
    nq = len(query_metas)
    pg = ctx["_packed"]
    dev = pg.lens.device
    pq = scoring.pack_queries(qs)
    ws = scoring.simpool_partials(pq, pg)
    ptr, idx, first, has = ddist.local_gt_csr(t2v_gt, nq, lo, hi)
    ptr_d, idx_d, first_d = (torch.from_numpy(a).to(dev) for a in (ptr, idx, first))
    bad = pq.bad[:nq] > 0
    thr_fn = lambda: scoring.shard_thresholds(ws, pq, pg, ptr_d, idx_d, first_d)        # noqa: E731
    cnt_fn = lambda thr: scoring.shard_counts(ws, pq, pg, thr)                          # noqa: E731
    if use_collectives:
        ranks = ddist.sharded_ranks_from_partials(thr_fn, cnt_fn, torch.from_numpy(has), bad, n_videos)
    else:                                                     # no communicator: the one shard is the gallery
        thr, flag = thr_fn()
        r = torch.clamp(cnt_fn(thr).to(torch.int64) + 1, max=n_videos + 1)
        worst = (flag > 0) | (~torch.from_numpy(has).to(dev) | bad)[None, None, :]
        ranks = torch.where(worst, torch.full_like(r, n_videos + 1), r)
    if guard and use_collectives:
        _comm.current().all_reduce(ops.nonfinite_flag(opt.device), "max")       # every rank takes the same branch below
    if ranks.is_cuda:
        c = _comm.current()
        if c is not None:
            c.host_wait(what="eval_epoch_sharded: all-reduces of thresholds and counts")    # deadline-bounded (comm.py)
    if guard and ops.take_nonfinite(opt.device):                                # fp16 overflow guard: see eval_epoch
        if _fp16_overflow_policy(opt) == "raise":
            raise RuntimeError(_OVERFLOW_TEXT)
        logger.warning(_OVERFLOW_TEXT + ": repeating the sharded evaluation in parity mode")
        return eval_epoch_sharded(model, val_video_dataset, val_text_dataset, _ParityOpt(opt), test)
    ranks = ranks.cpu().numpy()
    out = {}
    kinds = (("inher", 0), ("explore", 1), ("fused", 2)) if model.double_branch else (("inher", 0), ("fused", 0))
    for name, k in kinds:
        out[name] = _recalls(ranks[k, 0], nq)
        logging.info(" * %s r_1_5_10_100: %s", name, [round(x, 1) for x in out[name][:4]])
    r1, r5, r10, r100 = out["fused"][:4]
    return r1 + r5 + r10 + r100


# Precision of eval_epoch / eval_epoch_sharded (opt.eval_precision overrides; test=True evaluates in "parity" unless overridden):
#   "throughput" (default; BASELINE configs[1] is bf16): bf16 input projection K4 + the fused bf16 tower kernel K5 + the bf16 scorer.
#       Gated at R@1/5/10/100 within 0.1 of the fp32 CPU restatement from raw features (tests/test_rk_gate_gpu.py, tools/rk_gate.py).
#   "parity": whatever ops.gemm_precision() is set to - by default the fp32-grade towers every golden test runs on.
EVAL_PRECISION = "throughput"


def eval_precision_mode(opt, test=False):
    """opt.eval_precision when given; otherwise the gated throughput mode for per-epoch validation and 'parity' for the final / test
    evaluation (test=True): the numbers a run reports as its result are computed the way the reference computes them (fp32-grade),
    the bf16 path - within 0.1 of it at every cut, tests/test_rk_gate_gpu.py - serves model selection during training."""
    return getattr(opt, "eval_precision", None) or ("parity" if test else EVAL_PRECISION)


@contextlib.contextmanager
def eval_precision(model, opt, test=False):
    from . import ops
    mode = eval_precision_mode(opt, test)
    logger.info(f"evaluation precision: {mode}" + (" (bf16 input projection + fused bf16 towers + bf16 scorer; R@K within 0.1 of fp32: "
                                                   "profiles/r04/rk_gate.json)" if mode == "throughput" else " (fp32-grade towers)"))
    if mode not in ("throughput", "parity"):
        raise ValueError(f"eval_precision must be 'throughput' or 'parity', got {mode!r}")
    if mode == "parity":
        if ops.precision_mode() in ("mixed", "fp32x2"):      # training precisions whose forward is not the three-plane parity grade
            prev_mode = ops.precision_mode()
            ops.set_gemm_precision("fp32")
            try:
                yield
            finally:
                ops.set_gemm_precision(prev_mode)
            return
        yield
        return
    prev = (ops.precision_mode(), model.fast_input_proj)
    ops.set_gemm_precision("bf16")
    model.fast_input_proj = True
    try:
        yield
    finally:
        ops.set_gemm_precision(prev[0])
        model.fast_input_proj = prev[1]


def _fp16_overflow_policy(opt):
    """opt.eval_overflow: "parity" (default) re-runs the evaluation in parity mode, "raise" raises RuntimeError."""
    pol = getattr(opt, "eval_overflow", None) or "parity"
    if pol not in ("parity", "raise"):
        raise ValueError(f"eval_overflow must be 'parity' or 'raise', got {pol!r}")
    return pol


_OVERFLOW_TEXT = ("throughput-mode evaluation: an activation of this checkpoint overflowed the fp16 operands of the fused towers "
                  "(|x| > 65,504: the towers' second LayerNorm met a non-finite row)")


class _ParityOpt:
    """`opt` with eval_precision forced to "parity" (attribute reads fall through to the caller's object)."""

    def __init__(self, opt):
        object.__setattr__(self, "_opt", opt)

    def __getattr__(self, k):
        if k == "eval_precision":
            return "parity"
        return getattr(object.__getattribute__(self, "_opt"), k)

    def __setattr__(self, k, v):
        setattr(object.__getattribute__(self, "_opt"), k, v)


def eval_epoch(model, val_video_dataset, val_text_dataset, opt, test=False):
    """SumR (R@1 + R@5 + R@10 + R@100) of the fused scores; logs the three rankings (eval.py:237-263).

    fp16 overflow guard (throughput mode): the fused towers raise a device flag when a valid row turns non-finite (an activation
    beyond fp16's 65,504: ops.nonfinite_flag); it is read here, after the epoch's results (no extra synchronisation on the normal
    path), and the evaluation is repeated in parity mode - or RuntimeError with opt.eval_overflow = "raise" - instead of ranking
    NaN scores last in silence (the reference's fp32 path has no such cliff: model_components.py:305-312,398-436)."""
    from . import ops
    with host_threads():
        mode = eval_precision_mode(opt, test)
        dev = torch.device(getattr(opt, "device", "cuda"))
        guard = mode == "throughput" and dev.type == "cuda"
        if guard:
            ops.nonfinite_flag(dev).zero_()
        with eval_precision(model, opt, test):
            res = _eval_epoch(model, val_video_dataset, val_text_dataset, opt, test)
        if guard and ops.take_nonfinite(dev):
            if _fp16_overflow_policy(opt) == "raise":
                raise RuntimeError(_OVERFLOW_TEXT)
            logger.warning(_OVERFLOW_TEXT + ": repeating the evaluation in parity mode")
            popt = _ParityOpt(opt)
            with eval_precision(model, popt, test):
                res = _eval_epoch(model, val_video_dataset, val_text_dataset, popt, test)
        return res


_GT_CACHE = {}     # (videos, captions) -> (t2v_gt, ptr, idx): the validation sets do not change between epochs


def _gt_cached(video_metas, query_metas, device):
    """get_gt + gt_csr of one (gallery, caption set) pair, kept across epochs (13 + 7 ms of Python per epoch at TVR size -
    a third of a cached eval_epoch's wall time, profiles/r06/eval_epoch_c2_cached_cprofile.txt); keyed by the ids themselves."""
    key = (tuple(video_metas), tuple(query_metas), str(device))
    hit = _GT_CACHE.get(key)
    if hit is None:
        _, t2v_gt = get_gt(video_metas, query_metas)
        ptr, idx = gt_csr(t2v_gt, len(query_metas), device)
        if len(_GT_CACHE) >= 4:
            _GT_CACHE.pop(next(iter(_GT_CACHE)))
        hit = _GT_CACHE[key] = (t2v_gt, ptr, idx)
    return hit


def rank_queries(model, eval_dataset, opt, ctx_info, w=(0.7, 0.3)):
    """What eval_epoch needs of compute_query2ctx_info + the three cal_perf calls, without the score matrices: encode the
    queries, score them against the resident gallery (scorer partial planes only), and rank the ground-truth videos straight
    from the planes (scoring.rank_partials).  Returns (ranks (3, 2, Nq) int32 numpy, query_metas)."""
    model.eval()
    metas, qs = _encode_all_queries(model, eval_dataset, opt)
    pg = ctx_info["_packed"]
    pq = scoring.pack_queries(qs)
    ws = scoring.simpool_partials(pq, pg)
    _, ptr, idx = _gt_cached(ctx_info["video_metas"], metas, pg.lens.device)
    return scoring.rank_partials(ws, pq, pg, ptr, idx, w).cpu().numpy(), metas


def _eval_epoch(model, val_video_dataset, val_text_dataset, opt, test=False):
    model.eval()
    logger.info("Computing scores")
    context_info = compute_context_info(model, val_video_dataset, opt, keep_frame_feats=False)
    ranks, query_metas = rank_queries(model, val_text_dataset, opt, context_info)
    nq = len(query_metas)
    if opt.double_branch:
        logging.info("inher_scores:")
        perf_from_ranks(ranks[0, 0], ranks[0, 1], nq)
        logging.info("explore_scores:")
        perf_from_ranks(ranks[1, 0], ranks[1, 1], nq)
        logging.info("score_sum:")
        r1, r5, r10, r100, _, _, _ = perf_from_ranks(ranks[2, 0], ranks[2, 1], nq)
    else:
        r1, r5, r10, r100, _, _, _ = perf_from_ranks(ranks[0, 0], ranks[0, 1], nq)
    return r1 + r5 + r10 + r100
