"""Host side of the scoring path (K1 `simpool`): packing and launching.

Mirrors what DLDKD.get_sim_scores + compute_query2ctx_info + eval_epoch's fusion do in the reference
(method/model.py:307-329, method/eval.py:200-208,254) but keeps the gallery resident in a packed bf16
layout and never builds the (Nq, L, Nv) clip tensor.
"""
import torch

from . import native

HIDDEN = 384
MAX_CLIPS = 128


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class PackedQueries:
    """bf16 queries in MFMA B-fragment order, one blob per branch."""

    def __init__(self, blobs, nq):
        self.blobs = blobs
        self.nq = nq


class PackedGallery:
    """Resident bf16 gallery: one blob per branch + lens + a visiting order (descending length)."""

    def __init__(self, blobs, lens, order, inv_order, nv, L):
        self.blobs, self.lens, self.order, self.inv_order, self.nv, self.L = blobs, lens, order, inv_order, nv, L

    @property
    def n_branches(self):
        return len(self.blobs)


def pack_queries(qs, normalize=True):
    """qs: list (one per branch) of (Nq, 384) tensors on the GPU."""
    L = native.lib()
    nq = qs[0].shape[0]
    blobs = []
    for q in qs:
        if q.dim() != 2 or q.shape[1] != HIDDEN or q.shape[0] != nq:
            raise native.NativeError(f"queries must be (Nq, {HIDDEN}); got {tuple(q.shape)}")
        q = _f32c(q)
        blob = torch.empty(L.dldkd_packed_queries_bytes(nq), dtype=torch.uint8, device=q.device)
        native.check(L.dldkd_pack_queries_bf16(native.ptr(q), nq, int(normalize), native.ptr(blob), native.stream()),
                     "pack_queries")
        blobs.append(blob)
    return PackedQueries(blobs, nq)


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
    # longest first: the 4 waves of a workgroup get similar lengths and the tail of the grid is light
    order = torch.argsort(lens, descending=True, stable=True).to(torch.int32)
    inv = torch.empty_like(order)
    inv[order.long()] = torch.arange(nv, dtype=torch.int32, device=dev)
    return PackedGallery(blobs, lens, order, inv, nv, L)


class GalleryPacker:
    """Streaming pack_gallery: the eval driver hands over one encoded batch at a time and the fp32 (Nv, L, 384)
    gallery of the reference (eval.py:139-175, 2 x 4.3 GB at TVR scale) never exists.  add() packs videos
    [v0, v0 + n) of every branch; finish() computes the visiting order and returns the PackedGallery."""

    def __init__(self, nv, L, n_branches, device, normalize=True):
        if L > MAX_CLIPS:
            raise native.NativeError(f"at most {MAX_CLIPS} clips per video (config max_ctx_l); got {L}")
        L_ = native.lib()
        self.nv, self.L, self.normalize, self.filled = nv, L, normalize, 0
        self.blobs = [torch.empty(L_.dldkd_packed_gallery_bytes(nv, L), dtype=torch.uint8, device=device)
                      for _ in range(n_branches)]
        self.lens = torch.zeros(max(nv, 1), dtype=torch.int32, device=device)

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
        order = torch.argsort(lens, descending=True, stable=True).to(torch.int32)
        inv = torch.empty_like(order)
        inv[order.long()] = torch.arange(self.nv, dtype=torch.int32, device=lens.device)
        return PackedGallery(self.blobs, lens, order, inv, self.nv, self.L)


def _variant():
    import os
    return os.environ.get("DLDKD_SIMPOOL_VARIANT", "2")


def _units(pg):
    """Half-video units of scorer v3, built once per gallery: (unit_video, unit_row0, unit_rows, video_unit0,
    video_unit1, n_units).  Units are visited in descending ceil(rows/16) order."""
    if getattr(pg, "_units", None) is None:
        lens = pg.lens.long()
        dev = lens.device
        vid = torch.arange(pg.nv, device=dev)
        two = lens > 64
        uv = torch.cat([vid, vid[two]])
        row0 = torch.cat([torch.zeros(pg.nv, dtype=torch.long, device=dev), torch.full((int(two.sum()),), 64, device=dev)])
        rows = torch.cat([torch.clamp(lens, max=64), lens[two] - 64])
        order = torch.argsort((rows + 15) // 16, descending=True, stable=True)
        uv, row0, rows = uv[order], row0[order], rows[order]
        n_units = uv.numel()
        pos = torch.empty(n_units, dtype=torch.long, device=dev)
        pos[order] = torch.arange(n_units, device=dev)
        u0 = pos[:pg.nv]
        u1 = torch.full((pg.nv,), -1, dtype=torch.long, device=dev)
        u1[two] = pos[pg.nv:]
        pg._units = tuple(t.to(torch.int32).contiguous() for t in (uv, row0, rows, u0, u1)) + (n_units,)
    return pg._units


def _stream_plan(pg):
    """Row-stream plan of scorer v4, built once per gallery on the host (dldkd_simpool_plan_stream) and kept on the
    device: (rowsrc, tile_end, tile_unit, tail_unit, video_unit0, video_unit1, n_waves, n_units)."""
    if getattr(pg, "_stream", None) is None:
        import ctypes
        import numpy as np
        lens = np.ascontiguousarray(pg.lens.cpu().numpy().astype(np.int32))
        lp = (pg.L + 31) // 32 * 32
        max_waves = int((int(lens.sum()) + 15 * pg.nv) // 128 + 2)
        rowsrc = np.empty(max_waves * 128, np.int32)
        te, tu = np.empty(max_waves * 8, np.int32), np.empty(max_waves * 8, np.int32)
        tail = np.empty(max_waves, np.int32)
        u0, u1 = np.empty(max(pg.nv, 1), np.int32), np.empty(max(pg.nv, 1), np.int32)
        nw, nu = ctypes.c_int(0), ctypes.c_int(0)
        hp = lambda a: ctypes.c_void_p(a.ctypes.data)   # noqa: E731
        native.check(native.lib().dldkd_simpool_plan_stream(hp(lens), pg.nv, lp, max_waves, hp(rowsrc), hp(te), hp(tu), hp(tail),
                                                            hp(u0), hp(u1), ctypes.cast(ctypes.byref(nw), ctypes.c_void_p),
                                                            ctypes.cast(ctypes.byref(nu), ctypes.c_void_p)), "simpool_plan_stream")
        nw, nu = nw.value, nu.value
        dev = pg.lens.device
        up = lambda a: torch.from_numpy(a).to(dev)      # noqa: E731
        pg._stream = (up(rowsrc[:max(nw, 1) * 128]), up(te[:max(nw, 1) * 8]), up(tu[:max(nw, 1) * 8]), up(tail[:max(nw, 1)]),
                      up(u0), up(u1), nw, nu)
    return pg._stream


def simpool_partials(pq, pg, workspace=None):
    """Stage 1 (the dominant kernel): per-branch pooled scores into the workspace, transposed and in
    visiting order.  Returns the workspace tensor."""
    L_ = native.lib()
    nb = pg.n_branches
    if len(pq.blobs) != nb:
        raise native.NativeError("query / gallery branch count mismatch")
    if _variant() == "4":
        rowsrc, te, tu, tail, _, _, n_waves, n_units = _stream_plan(pg)
        need = L_.dldkd_simpool_units_workspace_bytes(pq.nq, n_units, nb)
        if workspace is None or workspace.numel() < need:
            workspace = torch.empty(need, dtype=torch.uint8, device=pg.lens.device)
        if pq.nq and pg.nv:
            native.check(L_.dldkd_simpool_eval_stream_bf16(native.ptr_array(pq.blobs), native.ptr_array(pg.blobs), native.ptr(rowsrc),
                                                           native.ptr(te), native.ptr(tu), native.ptr(tail), pq.nq, n_waves, n_units,
                                                           nb, native.ptr(workspace), native.stream()), "simpool_eval_stream")
        return workspace
    if _variant() == "3":
        uv, row0, rows, _, _, n_units = _units(pg)
        need = L_.dldkd_simpool_units_workspace_bytes(pq.nq, n_units, nb)
        if workspace is None or workspace.numel() < need:
            workspace = torch.empty(need, dtype=torch.uint8, device=pg.lens.device)
        if pq.nq and pg.nv:
            native.check(L_.dldkd_simpool_eval_units_bf16(native.ptr_array(pq.blobs), native.ptr_array(pg.blobs), native.ptr(uv),
                                                          native.ptr(row0), native.ptr(rows), pq.nq, n_units, pg.L, nb,
                                                          native.ptr(workspace), native.stream()), "simpool_eval_units")
        return workspace
    need = L_.dldkd_simpool_eval_workspace_bytes(pq.nq, pg.nv, nb)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=pg.lens.device)
    if pq.nq and pg.nv:
        native.check(L_.dldkd_simpool_eval_bf16(native.ptr_array(pq.blobs), native.ptr_array(pg.blobs), native.ptr(pg.lens),
                                                native.ptr(pg.order), pq.nq, pg.nv, pg.L, nb, native.ptr(workspace),
                                                native.stream()), "simpool_eval")
    return workspace


def simpool_finish(workspace, pq, pg, w=(0.7, 0.3), want_fused=True, want_branches=False):
    """Stage 2: (Nq, Nv) outputs.  Returns (fused, s0, s1), each fp32 (Nq, Nv) or None."""
    L_ = native.lib()
    nb, nq, nv, dev = pg.n_branches, pq.nq, pg.nv, pg.lens.device
    fused = torch.empty(nq, nv, dtype=torch.float32, device=dev) if want_fused else None
    s0 = torch.empty(nq, nv, dtype=torch.float32, device=dev) if want_branches else None
    s1 = torch.empty(nq, nv, dtype=torch.float32, device=dev) if (want_branches and nb == 2) else None
    if nq and nv and _variant() in ("3", "4"):
        if _variant() == "4":
            _, _, _, _, u0, u1, _, n_units = _stream_plan(pg)
        else:
            _, _, _, u0, u1, n_units = _units(pg)
        native.check(L_.dldkd_simpool_finish_units(native.ptr(workspace), native.ptr(u0), native.ptr(u1), nq, nv, n_units, nb,
                                                   float(w[0]), float(w[1]), native.ptr(fused), native.ptr(s0), native.ptr(s1),
                                                   native.stream()), "simpool_finish_units")
    elif nq and nv:
        native.check(L_.dldkd_simpool_finish(native.ptr(workspace), native.ptr(pg.inv_order), nq, nv, nb, float(w[0]),
                                             float(w[1]), native.ptr(fused), native.ptr(s0), native.ptr(s1),
                                             native.stream()), "simpool_finish")
    return fused, s0, s1


def simpool_eval(pq, pg, w=(0.7, 0.3), want_fused=True, want_branches=False, workspace=None):
    """Pooled cosine/dot scores of every query against every video.

    Returns (fused, s0, s1): (Nq, Nv) fp32 tensors or None.  fused = w[0]*s0 + w[1]*s1 (eval.py:254),
    or s0 alone for a single-branch model.
    """
    ws = simpool_partials(pq, pg, workspace)
    return simpool_finish(ws, pq, pg, w, want_fused, want_branches)
