"""Batch assembly for the eval driver: the input contract of the hot path (padding + 0/1 masks).

Restates the two eval collates of the reference (method/data_provider.py:75-86,139-170); dataset file
readers (BigFile / HDF5) are out of scope - any Dataset yielding (feat (len, D) float32, index, id) works.
"""
import contextlib
import types

import numpy as np

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


# ------------------------------------------------------------------------------------------------------------------------
# Device-resident training set (the callers of the hot path).  The reference reads, pads and uploads every
# training item again in every epoch (DataLoader -> collate_train -> .to(device): ~200 MB of padded clips per TVR batch over
# PCIe, after a host-side pad of 128 ragged videos), although the items never change.  Here every item is read ONCE into ragged
# row tables on the device (TVR: 17,435 videos x <= 128 clips x 3072 fp32 = 27 GB of 288), and a batch is a gather from those
# tables by a kernel: the tensors collate_train + .to(device) would have produced, bit for bit, in the same order.
def _identity(x):
    return x


def _collate_train_tables(items):
    """DeviceTrainSet's reading pass in a WORKER process: the batch as {table: (rows (sum len, D) fp32, [len, ...])} + the items'
    caption counts - the concatenation runs in the worker, five tensors cross the process boundary."""
    out, caps_n = {}, []
    seqs = {k: [] for k in DeviceTrainSet.TABLES}
    for it in items:
        s_vid, caps, t_vid, t_caps = it[0], it[1], it[2], it[3]
        if len(caps) != len(t_caps):
            raise ValueError("DeviceTrainSet: an item's caption and teacher-caption lists differ in length")
        seqs["student_videos"].append(s_vid)
        seqs["teacher_videos"].append(t_vid)
        seqs["student_text"].extend(caps)
        seqs["teacher_text"].extend(t_caps)
        caps_n.append(len(caps))
    for k, v in seqs.items():
        v = [torch.as_tensor(a, dtype=torch.float32) for a in v]
        if any(a.dim() != 2 for a in v):
            raise ValueError(f"DeviceTrainSet: {k} sequences must be (length, features)")
        out[k] = (torch.cat(v, 0) if v else torch.zeros(0, 0), [int(a.shape[0]) for a in v])
    out["caps"] = caps_n
    return out


class _PinnedAppender:
    """Rows of (length, D) float32 sequences appended into a ring of two pinned buffers; a full buffer is uploaded asynchronously
    into a device tensor of exactly its rows (appended to `chunks`) while the other one fills."""

    def __init__(self, device, chunks, ring_bytes=64 << 20):
        self.device, self.chunks, self.ring_bytes = torch.device(device), chunks, int(ring_bytes)
        self.pinned = self.device.type == "cuda"
        self.bufs, self.done, self.k, self.rows, self.D = None, [None, None], 0, 0, None

    def _start(self, D):
        self.D = int(D)
        self.cap = max(self.ring_bytes // (4 * self.D), 1)
        self.bufs = [torch.empty(self.cap, self.D, dtype=torch.float32, pin_memory=self.pinned) for _ in range(2)]
        self.views = [b.numpy() for b in self.bufs]

    def add(self, a):
        if self.D is None:
            self._start(a.shape[1])
        if a.shape[1] != self.D:
            raise ValueError(f"DeviceTrainSet: feature width {a.shape[1]} in a table of width {self.D}")
        n, pos = int(a.shape[0]), 0
        while pos < n:                          # (a sequence longer than what is left of the buffer continues in the next one)
            if self.rows == self.cap:
                self.flush()
            m = min(n - pos, self.cap - self.rows)
            # (numpy's single-threaded memcpy: torch's copy_ spreads a 1-MB copy over every core of the host - 0.3 GB/s on a
            # 256-thread box, 25k sequences in 6 s)
            dst = self.views[self.k][self.rows:self.rows + m]
            if a.device.type == "cpu":
                np.copyto(dst, a[pos:pos + m].detach().numpy(), casting="unsafe")
            else:
                self.bufs[self.k][self.rows:self.rows + m].copy_(a[pos:pos + m])
            self.rows += m
            pos += m

    def flush(self, final=False):
        if self.rows:
            out = torch.empty(self.rows, self.D, dtype=torch.float32, device=self.device)
            out.copy_(self.bufs[self.k][:self.rows], non_blocking=True)
            self.chunks.append(out)
            if self.pinned:
                self.done[self.k] = torch.cuda.Event()
                self.done[self.k].record()
            self.k ^= 1
            self.rows = 0
            if self.done[self.k] is not None:
                self.done[self.k].synchronize()          # the upload that last read the buffer about to be refilled
        if final:
            for e in self.done:
                if e is not None:
                    e.synchronize()
            self.bufs = self.views = None


class DeviceTrainSet:
    TABLES = ("student_videos", "teacher_videos", "student_text", "teacher_text")

    def __init__(self, dataset, device, num_workers=0, cap_gb=160.0):
        import numpy as np
        from torch.utils.data import DataLoader
        self.device = torch.device(device)
        rows = {k: [] for k in self.TABLES}
        lens = {k: [] for k in self.TABLES}
        self.caps_of = []                       # per video: (first caption, number of captions)
        # (with workers: every table's rows concatenated BY the worker - a handful of tensors per batch through shared memory, not
        # the ~800 an identity collate would pass)
        loader = DataLoader(dataset, batch_size=64, shuffle=False, num_workers=num_workers,
                            collate_fn=_collate_train_tables if num_workers else _identity)
        rng = torch.get_rng_state()             # the reading pass must not move the run's random stream (a DataLoader iterator
        try:                                    # draws its base seed from the global generator)
            chunks = list(self._read(loader, rows, lens, cap_gb))
        finally:
            torch.set_rng_state(rng)
        n_caps = chunks[-1] if chunks else 0
        self._finish(rows, lens, n_caps)

    def _read(self, loader, rows, lens, cap_gb):
        """Every item's sequences go through per-table pinned staging buffers (_PinnedAppender): one memcpy per sequence into memory
        that is touched once and reused, uploaded asynchronously while the next items are read - not a torch.cat into fresh pageable
        memory + a pageable upload per 64 items (7 of the 9.5 s a 2,048-video TVR-shaped set took to build)."""
        n_caps, nbytes = 0, 0
        app = {k: _PinnedAppender(self.device, rows[k]) for k in self.TABLES}
        for chunk in loader:                    # 64 items at a time
            if isinstance(chunk, dict):             # a worker's batch: per table (rows, lens) + the items' caption counts
                for k in self.TABLES:
                    rows_k, lens_k = chunk[k]
                    if lens_k:
                        app[k].add(rows_k)
                    lens[k].extend(lens_k)
                    nbytes += rows_k.numel() * 4
                for c in chunk["caps"]:
                    self.caps_of.append((n_caps, c))
                    n_caps += c
                chunk = ()
            for item in chunk:
                s_vid, caps, t_vid, t_caps = item[0], item[1], item[2], item[3]
                for k, seqs in (("student_videos", [s_vid]), ("teacher_videos", [t_vid]), ("student_text", caps), ("teacher_text", t_caps)):
                    for a in seqs:
                        a = torch.as_tensor(a)
                        if a.dim() != 2:
                            raise ValueError(f"DeviceTrainSet: {k} sequences must be (length, features); got {tuple(a.shape)}")
                        app[k].add(a)
                        lens[k].append(int(a.shape[0]))
                        nbytes += a.numel() * 4
                if len(caps) != len(t_caps):
                    raise ValueError("DeviceTrainSet: an item's caption and teacher-caption lists differ in length")
                self.caps_of.append((n_caps, len(caps)))
                n_caps += len(caps)
            if nbytes > cap_gb * 1e9:
                raise MemoryError(f"DeviceTrainSet: the training set exceeds the {cap_gb} GB cap")
            yield n_caps
        for a in app.values():
            a.flush(final=True)

    def _finish(self, rows, lens, n_caps):
        import numpy as np
        self.n_videos, self.n_caps = len(self.caps_of), n_caps
        self.lens_host = {k: np.asarray(v, dtype=np.int32) for k, v in lens.items()}
        self.src, self.row_start, self.lens_dev, self.dim = {}, {}, {}, {}
        for k in self.TABLES:
            if not rows[k]:
                raise ValueError("DeviceTrainSet: empty dataset")
            self.dim[k] = int(rows[k][0].shape[1])
            if self.dim[k] % 4:
                raise ValueError("DeviceTrainSet: feature widths must be multiples of 4")
            self.src[k] = rows[k][0] if len(rows[k]) == 1 else torch.cat(rows[k], 0)
            start = np.zeros(len(lens[k]), dtype=np.int64)
            start[1:] = np.cumsum(self.lens_host[k][:-1], dtype=np.int64)
            self.row_start[k] = torch.from_numpy(start).to(self.device)
            self.lens_dev[k] = torch.from_numpy(self.lens_host[k]).to(self.device)
            rows[k] = None
        self.n_caps_of = np.asarray([c for _, c in self.caps_of], dtype=np.int64)
        self._ring = None

    def __len__(self):
        return self.n_videos

    def _gather(self, k, items_dev, n, lmax, out=None, mask=None, rows=None):
        """rows >= n: the destination holds `rows` sequences; those behind the n gathered ones become padding sequences - zero
        features, ONE valid position (mask[:, 0] = 1: a tower never sees an empty sequence)."""
        from . import native
        D = self.dim[k]
        rows = n if rows is None else int(rows)
        if out is None:
            out = torch.empty(rows, lmax, D, dtype=torch.float32, device=self.device)
        if mask is None:
            mask = torch.empty(rows, lmax, dtype=torch.float32, device=self.device)
        if (rows < n or tuple(out.shape) != (rows, lmax, D) or tuple(mask.shape) != (rows, lmax)
                or not (out.is_contiguous() and mask.is_contiguous())):
            raise ValueError(f"DeviceTrainSet: destination of {k} must be contiguous ({rows}, {lmax}, {D}) + ({rows}, {lmax})")
        native.check(native.lib().dldkd_gather_pad_rows_f32(native.ptr(self.src[k]), native.ptr(self.row_start[k]), native.ptr(self.lens_dev[k]),
                                                            native.ptr(items_dev), n, lmax, D, native.ptr(out), native.ptr(mask),
                                                            native.stream()), "gather_pad_rows")
        if rows > n:
            out[n:].zero_()
            mask[n:].zero_()
            mask[n:, 0] = 1.0
        return out, mask

    def plan(self, indices):
        """Host half of a batch: the items in collate order, the captions, the labels and every table's longest sequence - all a
        consumer needs to know the batch's shapes before a single row has moved (train.GraphedTrainStep.iterate picks the input
        buffers of the captured step from them and has the rows gathered straight into those)."""
        import numpy as np
        idx = np.asarray(indices, dtype=np.int64)
        order = np.argsort(-self.n_caps_of[idx], kind="stable")        # most captions first, ties in sampler order (sorted() is stable)
        vids = idx[order]
        caps = np.concatenate([np.arange(self.caps_of[v][0], self.caps_of[v][0] + self.caps_of[v][1]) for v in vids])
        labels = [vi for vi, v in enumerate(vids) for _ in range(self.caps_of[v][1])]
        items = {"student_videos": vids, "teacher_videos": vids, "student_text": caps, "teacher_text": caps}
        lmax = {k: int(self.lens_host[k][items[k]].max()) for k in self.TABLES}
        return types.SimpleNamespace(vids=vids, caps=caps, labels=labels, lmax=lmax, n={k: len(items[k]) for k in self.TABLES},
                                     dim=dict(self.dim))

    def gather(self, plan, out=None, pad=None, n_queries=None):
        """Device half: collate_train's dict for `plan`, gathered on the current stream.  pad: {table: padded length >= its longest
        sequence} (zero rows, zero mask behind a sequence's end: what the stepper's bucketing would append); n_queries >= the
        plan's captions: the text tensors get that many rows, the extra ones as padding queries (train.GraphedTrainStep pads the
        query axis to a bucket; text_labels stays the real list); out: a dict of destination tensors of exactly those shapes (the
        captured step's input buffers) - the rows then land where the step reads them and the returned dict holds those tensors."""
        from .staging import PinnedRing
        import numpy as np
        pad = pad or {}
        L = {k: max(int(pad.get(k, 0)), plan.lmax[k]) for k in self.TABLES}
        # one upload for both index lists (pinned slot read by a kernel: staging.PinnedRing / dldkd_upload_words)
        both = np.concatenate([plan.vids, plan.caps]).astype(np.int32)
        if self._ring is None or self._ring.bufs[0].numel() < both.nbytes:
            self._ring = PinnedRing(max(both.nbytes, 1 << 16), self.device, slots=8)
        slot = self._ring.next()
        slot[:both.nbytes].view(torch.int32).copy_(torch.from_numpy(both))
        dev = torch.empty(both.nbytes, dtype=torch.uint8, device=self.device)
        self._ring.upload(dev, by_kernel=True)
        dev = dev.view(torch.int32)
        v_dev, c_dev = dev[:len(plan.vids)], dev[len(plan.vids):]
        o = out or {}
        nv, nq = len(plan.vids), len(plan.caps)
        sv, sm = self._gather("student_videos", v_dev, nv, L["student_videos"], o.get("student_videos"), o.get("student_videos_mask"))
        tv, _ = self._gather("teacher_videos", v_dev, nv, L["teacher_videos"], o.get("teacher_videos"))
        st, stm = self._gather("student_text", c_dev, nq, L["student_text"], o.get("student_text"), o.get("student_text_mask"), rows=n_queries)
        tt, _ = self._gather("teacher_text", c_dev, nq, L["teacher_text"], o.get("teacher_text"), rows=n_queries)
        return dict(student_videos=sv, teacher_videos=tv, student_videos_mask=sm, student_text=st, student_text_mask=stm,
                    teacher_text=tt, text_labels=plan.labels)

    def batch(self, indices):
        """The training batch of dataset items `indices` (in sampler order): collate_train's dict with device tensors."""
        return self.gather(self.plan(indices))


class DeviceTrainLoader:
    """Iterates DeviceTrainSet batches in the order - and with the random draws - of the DataLoader it stands in for: an index-only
    DataLoader with the same batch size / shuffle / sampler produces the index lists (same base-seed and permutation draws from the
    global generator), the batches themselves are gathered on the device."""

    def __init__(self, devset, batch_size, shuffle=True, sampler=None):
        from torch.utils.data import DataLoader
        self.devset, self.sampler = devset, sampler
        self._index_loader = DataLoader(range(len(devset)), batch_size=batch_size, shuffle=shuffle and sampler is None, sampler=sampler,
                                        num_workers=0, collate_fn=lambda x: [int(i) for i in x])

    def __len__(self):
        return len(self._index_loader)

    def __iter__(self):
        for idx in self._index_loader:
            yield self.devset.batch(idx)

    def plans(self):
        """The epoch's batches as host-side plans (DeviceTrainSet.plan), in the order and with the random draws of __iter__; the
        consumer gathers each with devset.gather(plan, out=..., pad=...)."""
        for idx in self._index_loader:
            yield self.devset.plan(idx)
