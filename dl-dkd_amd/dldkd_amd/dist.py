"""One-process-per-GPU pieces of the path.  The collectives go through a communicator (comm.py): on the GPU RCCL over xGMI driven
directly through the C ABI (comm.RcclComm: every collective one enqueue on the caller's stream, no process-group watchdog), in the
CPU tests a torch.distributed gloo group (comm.TorchGroupComm).  `comm=None` everywhere = comm.current().  The reference is
single-GPU; these are the two exchanges BASELINE.json's north_star adds.

Eval: the gallery shards by video (independent units).  Rank r keeps videos [r*S, (r+1)*S), S = ceil(Nv/N);
every rank scores ALL queries against its shard; then either
  * OverlappedShardScorer: all_gather of the (Nq_r, S) blocks, one query range at a time under the scoring of the next
    ranges -> (Nq, Nv) on every rank (north_star's exchange), or
  * sharded_gt_ranks(): no matrix exchange at all - all-reduce(MAX) of the ground-truth scores (Nq floats),
    local count of shard videos above them, all-reduce(SUM) of the counts (Nq ints): exact R@K.
Training: local in-batch losses (model.py:353-387 defines negatives within one batch), one flat fp32 gradient
buffer (the optimizer's own, optimization.FlatParams.grad) mean-all-reduced per step (sync_gradients)."""
import torch

from . import comm as _comm


def _c(comm):
    c = comm if comm is not None else _comm.current()
    if c is None:
        raise RuntimeError("no communicator: comm.init_rccl_from_env(device) (GPU) or an initialised torch.distributed group (CPU tests)")
    return c


def shard_range(n_videos, rank, world):
    """(lo, hi, shard_size): contiguous ceil-division shards; the last ranks may be short or empty."""
    s = (n_videos + world - 1) // world
    lo = min(rank * s, n_videos)
    return lo, min(lo + s, n_videos), s


def _count_above_hip(scores, thr, n_valid):
    from . import native
    counts = torch.empty(scores.shape[0], dtype=torch.int32, device=scores.device)
    native.check(native.lib().dldkd_count_above_f32(native.ptr(scores), native.ptr(thr), scores.shape[0], n_valid,
                                                    scores.shape[1], native.ptr(counts), native.stream()), "count_above")
    return counts


def sharded_gt_ranks(local_scores, gt_video, n_videos, comm=None, count_fn=None):
    """Gather-free ranking.  local_scores (Nq, S) for this rank's shard; gt_video (Nq,) global index of each
    query's ground-truth video (-1: none).  Returns rank (Nq,) int64 = 1 + #videos scoring above the GT video, identical
    on every rank; n_videos + 1 for a query without ground truth or with a NaN ground-truth score (rank.hip NaN policy)."""
    c = _c(comm)
    rank, world = c.rank, c.world
    lo, hi, s = shard_range(n_videos, rank, world)
    nq = local_scores.shape[0]
    gt_video = gt_video.to(local_scores.device).long()
    mine = (gt_video >= lo) & (gt_video < hi)
    # row 0: the ground-truth score (-inf where it is not ours), row 1: 1 where it is NaN; one MAX all-reduce for both
    thr = torch.full((2, nq), float("-inf"), dtype=torch.float32, device=local_scores.device)
    thr[1].zero_()
    idx = torch.nonzero(mine).squeeze(1)
    if idx.numel():
        g = local_scores[idx, gt_video[idx] - lo].float()
        bad = torch.isnan(g)
        thr[0, idx] = torch.where(bad, torch.full_like(g, float("-inf")), g)
        thr[1, idx] = bad.float()
    c.all_reduce(thr, "max")
    count_fn = count_fn or _count_above_hip
    if hi > lo:
        counts = count_fn(local_scores.contiguous(), thr[0].contiguous(), hi - lo).to(torch.int64)
    else:                                                  # empty shard (world > n_videos): nothing to count
        counts = torch.zeros(nq, dtype=torch.int64, device=local_scores.device)
    c.all_reduce(counts, "sum")
    ranks = counts + 1
    worst = (thr[1] > 0) | (gt_video < 0)
    return torch.where(worst, torch.full_like(ranks, n_videos + 1), torch.clamp(ranks, max=n_videos + 1))


def local_gt_csr(t2v_gt, nq, lo, hi):
    """Ground truth cut to the shard [lo, hi): (ptr int32 (nq + 1), idx int32 local video indices, first_local int32 (nq,),
    has_gt bool (nq,)) as numpy arrays."""
    import numpy as np
    ptr = np.zeros(nq + 1, np.int32)
    idx, first, has = [], np.zeros(nq, np.int32), np.zeros(nq, bool)
    for q in range(nq):
        g = t2v_gt.get(q, []) if isinstance(t2v_gt, dict) else t2v_gt[q]
        if len(g):
            has[q] = True
            first[q] = int(lo <= g[0] < hi)
            idx.extend(v - lo for v in g if lo <= v < hi)
        ptr[q + 1] = len(idx)
    return ptr, np.asarray(idx if idx else [0], np.int32), first, has


def sharded_ranks_from_partials(local_thr_fn, local_count_fn, has_gt, bad, n_videos, comm=None):
    """Exact ranks of the ground-truth videos with the gallery sharded by video and NO score matrix anywhere: every rank computes
    thresholds over its own GT videos from its scorer's partial planes (local_thr_fn() -> (thr, nan_flag) fp32 (3, 2, Nq), -inf / 0
    where it holds none), all-reduce(MAX); counts its videos above them (local_count_fn(thr) -> int (3, 2, Nq)), all-reduce(SUM).
    has_gt (Nq,) bool and bad (Nq,) bool (NaN / Inf query vector) are the same on every rank.  Returns int64 (3, 2, Nq):
    [branch 0 / branch 1 / fused][best GT / first GT]; n_videos + 1 where there is no ground truth, the query is flagged, or the
    first GT video's score is NaN (rank.hip's NaN policy)."""
    c = _c(comm)
    thr, flag = local_thr_fn()
    both = torch.stack([thr, flag])                         # one collective for both
    c.all_reduce(both, "max")
    thr, flag = both[0], both[1]
    counts = local_count_fn(thr).to(torch.int64).contiguous()
    c.all_reduce(counts, "sum")
    ranks = torch.clamp(counts + 1, max=n_videos + 1)
    worst = (flag > 0) | (~has_gt.to(ranks.device) | bad.to(ranks.device))[None, None, :]
    return torch.where(worst, torch.full_like(ranks, n_videos + 1), ranks)


def all_reduce_flat(flat, comm=None):
    """Mean all-reduce of a flat gradient buffer (BertAdam's FlatParams.grad: every parameter's gradient is a view of ONE
    fp32 buffer, 23.0 MB for the TVR model / 17.5 MB for ActivityNet and Charades): one collective per step."""
    c = _c(comm)
    c.all_reduce(flat, "sum")
    flat.div_(c.world)


def sync_gradients(fp, comm=None, comm_stream=None, had=None):
    """The collective half of the data-parallel step (what DDP adds to method/train.py:141-151): gather whatever autograd left
    in p.grad into the flat buffer (parameters without a gradient contribute zeros), mean all-reduce it, and leave every
    p.grad pointing at its slice.  `fp` is an optimization.FlatParams; pure torch + the communicator, so it runs on CPU tensors
    with gloo.  had: the "this parameter had a gradient" flags when the gradients are already in the flat buffer (a replayed
    graph filled it: train.GraphedTrainStep) - nothing is gathered then.

    The all-reduce is ONE enqueue on the current stream (comm.RcclComm), ordered like any launch: between two graph replays of
    the step it needs no fence.  comm_stream: issue it from that stream instead, fenced both ways against the current one."""
    if had is None:
        had = fp.rebind_grads()
    c = _c(comm)
    check_had_flags(fp, had, c)
    if comm_stream is not None:
        cur = torch.cuda.current_stream(fp.grad.device)
        comm_stream.wait_stream(cur)
        with torch.cuda.stream(comm_stream):
            all_reduce_flat(fp.grad, c)
        cur.wait_stream(comm_stream)
    else:
        all_reduce_flat(fp.grad, c)
    # The "this parameter had no gradient" flags stay as they are: every rank builds the same autograd graph, so they agree
    # across ranks, and a parameter without a gradient on any rank must be SKIPPED by the optimizer (no weight decay, no moment
    # decay: optimization.py:294-295), exactly as on one GPU.


HAD_CHECK_EVERY = 256


def check_had_flags(fp, had, comm=None):
    """Every rank must hold a gradient for the SAME parameters: the optimizer skips a parameter without one (no weight decay, no
    moment decay: optimization.py:294-295); if the sets differed between ranks one replica would decay a parameter another skips
    and they would drift apart with no error.  One small SUM all-reduce of the 0 / 1 flags (each must come back 0 or
    world) and one host read.

    WHEN the check runs depends on nothing rank-local (a collective gated on this rank's own flag set pairs with
    another rank's gradient all-reduce): it is counted per optimizer - this function is called exactly once per data-parallel
    step on every rank - and runs on calls 0, 1 (the first eager and the first replayed step) and every HAD_CHECK_EVERY-th.
    Never issued under a graph capture (the collectives of the step are enqueued between the graphs, not inside them)."""
    n = getattr(fp, "_had_calls", 0)
    fp._had_calls = n + 1
    if not (n < 2 or n % HAD_CHECK_EVERY == 0):
        return
    if fp.grad.is_cuda and torch.cuda.is_current_stream_capturing():
        raise RuntimeError("check_had_flags under a graph capture: the data-parallel collectives belong between the graphs")
    c = _c(comm)
    mine = torch.tensor([1.0 if h else 0.0 for h in had], dtype=torch.float32, device=fp.grad.device)
    tot = mine.clone()
    c.all_reduce(tot, "sum")
    c.host_wait(what="check_had_flags all-reduce")
    tot = tot.cpu().tolist()
    bad = [i for i, t in enumerate(tot) if t != 0.0 and t != float(c.world)]
    if bad:
        raise RuntimeError(f"data parallel: the ranks disagree on which parameters have a gradient (parameter indices {bad[:8]}...): "
                           "the replicas would diverge")


class BucketedGradSync:
    """The same mean all-reduce, one collective per gradient bucket, each issued as soon as the backward pass has produced the
    bucket's last gradient (train.backward_in_phases calls bucket_ready) - the collective of a tower's gradients then runs
    while the next tower's backward kernels do (DDP's bucketed overlap, method/train.py:147-151 under DistributedDataParallel).
    The buckets are the contiguous ranges FlatParams laid out (optimization.FlatParams.bucket_ranges); a sum over ranks is
    element-wise, so the result is bit-identical to the single-bucket all-reduce.

    GPU: collectives go to `comm_stream` (so that they run beside the next tower's backward kernels), ordered after the gradient
    copies by an event and joined back in finish().  CPU / gloo: asynchronous work handles, waited in finish()."""

    def __init__(self, fp, comm=None, comm_stream=None):
        self.fp, self.comm, self.comm_stream = fp, _c(comm), comm_stream
        self.world = self.comm.world
        self.pending, self.issued = [], set()

    def _reduce(self, lo, hi):
        if hi <= lo:
            return
        part = self.fp.grad[lo:hi]
        if self.comm_stream is not None:
            cur = torch.cuda.current_stream(part.device)
            self.comm_stream.wait_stream(cur)
            with torch.cuda.stream(self.comm_stream):
                self.comm.all_reduce(part, "sum")
                part.div_(self.world)
        else:
            self.pending.append((self.comm.all_reduce(part, "sum", async_op=True), part))

    def bucket_ready(self, b):
        """Every gradient of bucket b is final: gather them into the flat range (eager) and start its all-reduce."""
        self.fp.rebind_bucket(b)
        self.issue(b)

    def issue(self, b):
        """Start the all-reduce of bucket b's range (the gradients are already in the flat buffer: replayed graph segment)."""
        if b not in self.issued:
            self.issued.add(b)
            self._reduce(*self.fp.bucket_ranges[b])

    def finish(self):
        """Whatever has not been issued goes now; then the current stream waits for every collective."""
        check_had_flags(self.fp, self.fp.rebind_grads(), self.comm)
        for b in range(len(self.fp.bucket_ranges)):
            self.issue(b)
        if self.comm_stream is not None:
            torch.cuda.current_stream(self.fp.grad.device).wait_stream(self.comm_stream)
        for work, part in self.pending:
            work.wait()
            part.div_(self.world)
        self.pending, self.issued = [], set()


def broadcast_parameters(fp, src=0, comm=None):
    """All replicas start from rank `src`'s parameters: one broadcast of the flat parameter buffer."""
    _c(comm).broadcast(fp.flat, src=src)


class ShardScorerBackend:
    """What OverlappedShardScorer drives (one implementation on HIP, one injected by the CPU tests):
        launch(done)            enqueue the scoring of ALL queries against the shard on the current stream; range r's
                                 scores are complete once done[r] == self.arrivals
        wait_range(done, r)      make the CURRENT stream wait for that (no host involvement)
        finish_range(r, lo, hi, out)   write the fused (hi - lo, shard) fp32 block of queries [lo, hi) into `out`"""
    arrivals = 0

    def launch(self, done):
        raise NotImplementedError

    def wait_range(self, done, r):
        raise NotImplementedError

    def finish_range(self, r, lo, hi, out):
        raise NotImplementedError


class HipShardBackend(ShardScorerBackend):
    """ONE launch of the MFMA scorer over all queries, grid [query range][branch][4 videos]: the ranges complete in order
    and every workgroup bumps its range's arrival counter after releasing its scores (simpool_eval.hip).  The consumer
    stream is parked on the counter with hipStreamWaitValue32 (dldkd_stream_wait_counter)."""

    def __init__(self, queries, gallery, min_ranges=4, w=(0.7, 0.3)):
        from . import native, scoring
        self.native, self.scoring = native, scoring
        self.queries, self.pg, self.w = queries, gallery, w
        # the range split is the KERNEL's (one planner, scoring.plan_query_split): a caller-made split that names more ranges
        # than the kernel creates would park the side stream on a counter nobody bumps
        nq = queries[0].shape[0]
        # (the grid is made of WAVES: with pair waves - two short videos per wave - fewer than one per video)
        waves = gallery.scorer_waves()
        self.n_ranges, self.per_range = scoring.plan_query_split(nq, waves, gallery.n_branches, min_split=min_ranges)
        self.bounds = [(lo, min(lo + self.per_range, nq)) for lo in range(0, max(nq, 1), self.per_range)]
        if len(self.bounds) != self.n_ranges or self.bounds[0][0] != 0 or self.bounds[-1][1] != nq:
            raise native.NativeError(f"HipShardBackend: {len(self.bounds)} ranges of {self.per_range} do not tile {nq} queries in "
                                     f"{self.n_ranges} kernel ranges")
        self.arrivals = (waves + 3) // 4 * gallery.n_branches
        self.ws = torch.empty(native.lib().dldkd_simpool_eval_workspace_bytes(queries[0].shape[0], gallery.nv, gallery.n_branches),
                              dtype=torch.uint8, device=gallery.lens.device)
        self.pq = None

    def launch(self, done):
        self.pq = self.scoring.pack_queries(self.queries)             # F.normalize + bf16 (model.py:318)
        self.scoring.simpool_partials(self.pq, self.pg, self.ws, q_split=self.n_ranges, done=done)

    def wait_range(self, done, r):
        native = self.native
        native.check(native.lib().dldkd_stream_wait_counter(native.stream(), native.ptr(done[r:]), self.arrivals), "stream_wait_counter")

    def finish_range(self, r, lo, hi, out):
        self.scoring.simpool_finish(self.ws, self.pq, self.pg, self.w, q_range=(lo, hi), out=out)


class OverlappedShardScorer:
    """One rank of the sharded all-pairs scoring step (the loop of method/eval.py:188-212 with the gallery cut by video):
    score all queries against this rank's shard in ONE launch whose query ranges complete in order; on a side stream, for every
    range as it completes: finish its (nq_r, S) block; on the collectives' stream, behind that block's event: all-gather it - the
    collective of range r runs under the scoring of ranges r+1.. (xGMI is point-to-point: the 7 peer transfers of one all-gather
    proceed in parallel).  Only the last range's gather is exposed.  Everything is enqueued up front; the host never waits.

    bounds: [(lo, hi)] query rows of every range; after step(): blocks[r] is (world * nq_r, S) rank-major;
    assemble() builds (Nq, n_videos)."""

    def __init__(self, backend, bounds, shard, device, comm=None, side_stream=None, comm_stream=None):
        bounds = list(bounds)
        if any(b[0] != a[1] for a, b in zip(bounds, bounds[1:])) or (bounds and bounds[0][0] != 0):
            raise ValueError(f"OverlappedShardScorer: ranges must tile the queries from 0 without gaps, got {bounds}")
        self.backend, self.bounds, self.shard, self.comm = backend, bounds, shard, _c(comm)
        self.world = self.comm.world
        self.local = [torch.empty(hi - lo, shard, dtype=torch.float32, device=device) for lo, hi in self.bounds]
        self.blocks = [torch.empty(self.world * (hi - lo), shard, dtype=torch.float32, device=device) for lo, hi in self.bounds]
        self.done = torch.zeros(max(len(self.bounds), 1), dtype=torch.int32, device=device)
        self.on_gpu = torch.device(device).type == "cuda"
        self.side = side_stream if (side_stream is not None or not self.on_gpu) else torch.cuda.Stream(device=device)
        # RCCL's kernels need CUs too, and the scorer parks a 512-register wave on every SIMD: the collectives' stream has high
        # priority, so its workgroups are dispatched first whenever a scorer workgroup retires
        self.comm_stream = comm_stream if (comm_stream is not None or not self.on_gpu) else torch.cuda.Stream(device=device, priority=-1)
        self.ready = [torch.cuda.Event() for _ in self.bounds] if self.on_gpu else []

    def step(self):
        if not self.on_gpu:                                   # CPU tests (gloo): same order of operations, no streams
            self.done.zero_()
            self.backend.launch(self.done)
            works = []
            for r, (lo, hi) in enumerate(self.bounds):
                self.backend.wait_range(self.done, r)
                self.backend.finish_range(r, lo, hi, self.local[r])
                works.append(self.comm.all_gather_into(self.blocks[r], self.local[r], async_op=True))
            for w in works:
                w.wait()
            return
        main = torch.cuda.current_stream()
        self.done.zero_()
        zeroed = torch.cuda.Event()
        zeroed.record(main)                                   # the side stream may look at the counters from here on...
        self.backend.launch(self.done)                        # ...while the scorer (main stream) is still running
        with torch.cuda.stream(self.side):
            self.side.wait_event(zeroed)                      # (also orders this step's blocks behind the last step's gathers)
            for r, (lo, hi) in enumerate(self.bounds):
                self.backend.wait_range(self.done, r)
                self.backend.finish_range(r, lo, hi, self.local[r])
                self.ready[r].record(self.side)
        with torch.cuda.stream(self.comm_stream):
            for r in range(len(self.bounds)):
                self.comm_stream.wait_event(self.ready[r])
                self.comm.all_gather_into(self.blocks[r], self.local[r])
        main.wait_stream(self.comm_stream)                    # (which has waited for every block of the side stream)

    def assemble(self, n_videos):
        rows = []
        for (lo, hi), blk in zip(self.bounds, self.blocks):
            n = hi - lo
            rows.append(blk.view(self.world, n, self.shard).permute(1, 0, 2).reshape(n, self.world * self.shard))
        return torch.cat(rows, 0)[:, :n_videos].contiguous()


def query_ranges(nq, n_ranges, per_range):
    """[(lo, hi)] of n_ranges ranges of per_range queries; raises if they do not cover [0, nq)."""
    out = [(lo, min(lo + per_range, nq)) for lo in range(0, max(nq, 1), per_range)]
    if len(out) != max(n_ranges, 1):
        raise ValueError(f"{n_ranges} ranges of {per_range} queries do not tile {nq} queries")
    return out
