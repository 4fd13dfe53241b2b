"""One-process-per-GPU pieces of the path (torch.distributed; backend "nccl" = RCCL over xGMI on MI355X,
"gloo" in the CPU tests).  The reference is single-GPU; these are the two exchanges
BASELINE.json's north_star adds.

Eval: the gallery shards by video (independent units).  Rank r keeps videos [r*S, (r+1)*S), S = ceil(Nv/N);
every rank scores ALL queries against its shard; then either
  * gather_scores(): one all_gather of the (Nq, S) blocks -> (Nq, Nv) on every rank (north_star's exchange), or
  * sharded_gt_ranks(): no matrix exchange at all - all-reduce(MAX) of the ground-truth scores (Nq floats),
    local count of shard videos above them, all-reduce(SUM) of the counts (Nq ints): exact R@K.
Training: local in-batch losses (model.py:353-387 defines negatives within one batch), one flat fp32 gradient
bucket all-reduced per step (FlatGradBucket)."""
import torch
import torch.distributed as dist


def shard_range(n_videos, rank, world):
    """(lo, hi, shard_size): contiguous ceil-division shards; the last ranks may be short or empty."""
    s = (n_videos + world - 1) // world
    lo = min(rank * s, n_videos)
    return lo, min(lo + s, n_videos), s


def gather_scores(local_scores, n_videos, group=None):
    """local_scores (Nq, S) (columns beyond the rank's real videos are padding) -> (Nq, n_videos)."""
    world = dist.get_world_size(group)
    nq, s = local_scores.shape
    out = torch.empty(world * nq, s, dtype=local_scores.dtype, device=local_scores.device)   # rank-major concatenation
    dist.all_gather_into_tensor(out, local_scores.contiguous(), group=group)
    return out.view(world, nq, s).permute(1, 0, 2).reshape(nq, world * s)[:, :n_videos].contiguous()


def _count_above_hip(scores, thr, n_valid):
    from . import native
    counts = torch.empty(scores.shape[0], dtype=torch.int32, device=scores.device)
    native.check(native.lib().dldkd_count_above_f32(native.ptr(scores), native.ptr(thr), scores.shape[0], n_valid,
                                                    scores.shape[1], native.ptr(counts), native.stream()), "count_above")
    return counts


def sharded_gt_ranks(local_scores, gt_video, n_videos, group=None, count_fn=None):
    """Gather-free ranking.  local_scores (Nq, S) for this rank's shard; gt_video (Nq,) global index of each
    query's ground-truth video.  Returns rank (Nq,) int64 = 1 + #videos scoring above the GT video, identical
    on every rank."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi, s = shard_range(n_videos, rank, world)
    nq = local_scores.shape[0]
    gt_video = gt_video.to(local_scores.device).long()
    mine = (gt_video >= lo) & (gt_video < hi)
    thr = torch.full((nq,), float("-inf"), dtype=torch.float32, device=local_scores.device)
    idx = torch.nonzero(mine).squeeze(1)
    thr[idx] = local_scores[idx, gt_video[idx] - lo].float()
    dist.all_reduce(thr, op=dist.ReduceOp.MAX, group=group)
    count_fn = count_fn or _count_above_hip
    counts = count_fn(local_scores.contiguous(), thr, hi - lo).to(torch.int64)
    dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts + 1


class FlatGradBucket:
    """All parameters' gradients as views of ONE flat fp32 buffer: the data-parallel step is a single
    all-reduce of 23.0 MB (TVR) / 17.5 MB (ActivityNet, Charades), no per-tensor launches or copies."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)
        self._bind()

    def _bind(self):
        o = 0
        for p in self.params:
            v = self.flat[o:o + p.numel()].view(p.shape)
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
            p.grad = v
            o += p.numel()

    def zero(self):
        self.flat.zero_()
        self._bind()

    def all_reduce_mean(self, group=None):
        self._bind()                       # autograd may have swapped in fresh .grad tensors
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.div_(dist.get_world_size(group))


def all_reduce_flat(flat, group=None):
    """Mean all-reduce of an existing flat gradient buffer (BertAdam's FlatParams.grad)."""
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(dist.get_world_size(group))


class OverlappedShardScorer:
    """Score all queries against this rank's gallery shard in `n_chunks` query chunks and all-gather each chunk's
    (nq_c, S) block asynchronously while the next chunk is being scored (the collective runs on RCCL's own stream;
    xGMI is point-to-point, so the 7 peer transfers of one all-gather proceed in parallel and hide under compute).

    score_chunk(lo, hi, out): writes the (hi - lo, S) fp32 block of queries [lo, hi) into `out`.
    After step(): self.blocks[c] is (world * nq_c, S) rank-major; assemble() builds (Nq, n_videos)."""

    def __init__(self, score_chunk, nq, shard, n_chunks, device, group=None):
        self.score_chunk, self.nq, self.shard, self.group = score_chunk, nq, shard, group
        self.world = dist.get_world_size(group)
        n_chunks = max(1, min(n_chunks, nq))
        step = (nq + n_chunks - 1) // n_chunks
        self.bounds = [(lo, min(lo + step, nq)) for lo in range(0, nq, step)]
        self.local = [torch.empty(hi - lo, shard, dtype=torch.float32, device=device) for lo, hi in self.bounds]
        self.blocks = [torch.empty(self.world * (hi - lo), shard, dtype=torch.float32, device=device) for lo, hi in self.bounds]

    def step(self):
        works = []
        for (lo, hi), loc, blk in zip(self.bounds, self.local, self.blocks):
            self.score_chunk(lo, hi, loc)
            works.append(dist.all_gather_into_tensor(blk, loc, group=self.group, async_op=True))
        for w in works:
            w.wait()

    def assemble(self, n_videos):
        rows = []
        for (lo, hi), blk in zip(self.bounds, self.blocks):
            n = hi - lo
            rows.append(blk.view(self.world, n, self.shard).permute(1, 0, 2).reshape(n, self.world * self.shard))
        return torch.cat(rows, 0)[:, :n_videos].contiguous()
