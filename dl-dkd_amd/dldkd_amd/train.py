"""Training driver: per-epoch scalar schedules, the optimisation step, best-checkpoint saving, early stop.

Mirrors reference method/train.py:52-247 for the part that touches the hot path; logging to
TensorBoard, result directories and code zips are out of scope.

Data parallel (BASELINE.json configs[4]; the reference is single-GPU): with torch.distributed initialised, one process per
GPU, every rank draws ITS OWN batches (DistributedSampler, reshuffled per epoch), starts from rank 0's parameters (one
broadcast of the flat buffer), computes local in-batch losses (the reference defines negatives within a batch,
model.py:353-387), and the step mean-all-reduces the one flat gradient buffer (dist.sync_gradients) before the fused
BertAdam update, which therefore stays identical on all ranks.  Dropout / triplet-sampling RNG is seeded rank-offset
(SURVEY 8e).  Rank 0 writes checkpoints; validation uses the gallery-sharded eval, whose SumR is identical on all ranks, so
early stopping is too."""
import logging
import math

import os

import torch
from torch.utils.data import DataLoader

from .data import collate_train
from .eval import eval_epoch
from .optimization import BertAdam

logger = logging.getLogger(__name__)


def _decay(kind, initial, floor, epoch_i, opt, sigmoid_k):
    """alpha / belta schedules (train.py:85-125)."""
    if kind == "exp":
        return max(initial * (opt.exponential_k ** epoch_i), floor)
    if kind == "linear":
        return max(initial + ((floor - initial) / opt.n_epoch) * epoch_i, floor)
    if kind == "sigmoid":
        return max(initial * (sigmoid_k / (sigmoid_k + math.exp(epoch_i * 100 / sigmoid_k))), floor)
    if kind == "cosine":
        return max(floor + 0.5 * (initial - floor) * (1 + math.cos(math.pi * epoch_i / opt.n_epoch)), floor)
    if kind == "None":
        return initial
    raise AssertionError(kind)


def epoch_schedules(opt, epoch_i):
    """(kd weight, alpha, belta) for an epoch; None where the reference leaves the attribute untouched."""
    weight = None
    d = getattr(opt, "distill_loss_decay", None)
    if d is not None:
        assert d in ["exp", "sigmoid", "linear", "None"]
        if d == "exp":
            weight = opt.exponential_k ** epoch_i                                           # train.py:76
        elif d == "linear":
            weight = max(opt.linear_k * epoch_i + opt.linear_b, 0.05)
        elif d == "sigmoid":
            weight = opt.sigmoid_k / (opt.sigmoid_k + math.exp(epoch_i * 100 / opt.sigmoid_k))
        else:
            weight = 1
    sk = opt.selfDistil_sigmoid_k
    alpha = belta = None
    if getattr(opt, "alpha_decay", None) is not None:
        assert opt.alpha_decay in ["exp", "sigmoid", "linear", "cosine", "None"]
        alpha = _decay(opt.alpha_decay, opt.alpha, 0.0, epoch_i, opt, sk)                   # min_alpha is 0 either way (:89-92)
    if getattr(opt, "belta_decay", None) is not None:
        assert opt.belta_decay in ["exp", "sigmoid", "linear", "cosine", "None"]
        belta = _decay(opt.belta_decay, opt.belta, 0 if opt.belta < 0.5 else 0.5, epoch_i, opt, sk)   # :109-113
    return weight, alpha, belta


def make_optimizer(model, opt, steps_per_epoch):
    """BertAdam over two groups: weight decay 0.01 except biases and LayerNorm parameters (train.py:203-213)."""
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    named = list(model.named_parameters())
    groups = [{"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
              {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
    # data parallel: ONE flat gradient range, one all-reduce behind the backward pass (a single replayed graph + a collective on the
    # comm stream).  opt.ddp_bucketed_overlap = True lays the flat buffers out tower by tower instead (one contiguous range per
    # tower, all-reduced while the next tower's backward pass runs: a chain of graph segments, _capture_segments): measured on a
    # one-rank RCCL group only, never against real RCCL kernels on two or more GPUs - off until a multi-GPU run has shown its
    # gradients and parameters equal the single-bucket run's over mixed eager / capture / replay steps.
    bucketed = bool(getattr(opt, "ddp_bucketed_overlap", False))
    buckets = model.grad_buckets() if (bucketed and dist_info()[1] >= DDP_MIN_WORLD and hasattr(model, "grad_buckets")) else None
    return BertAdam(groups, lr=opt.lr, weight_decay=opt.wd, warmup=opt.lr_warmup_proportion,
                    t_total=steps_per_epoch * opt.n_epoch, schedule="warmup_linear", grad_buckets=buckets)


DDP_MIN_WORLD = 2      # tests set 1 to drive the all-reduce branch with a one-rank group


def dist_info():
    """(rank, world) of the current communicator (comm.current(): the installed RCCL communicator, or the default torch.distributed
    group of the CPU tests), (0, 1) without one."""
    from . import comm
    return comm.info()


def backward_in_phases(loss, phases, after_phase=None, between=None):
    """loss.backward() cut at the tower outputs.  phases = [(tap, params)]: the loss reaches `params` only through the tensor
    `tap` (tap None: parameters the loss reaches without crossing any tap - their gradients come with the first pass).  First
    d loss / d taps, then one phase at a time d tap / d params with the tap's gradient fed in; after_phase(i) runs when phase
    i's gradients are final (the caller starts that bucket's all-reduce), between(i) before the next phase starts (the graph
    stepper closes one captured segment and opens the next there).  Per parameter these are the same kernels in the same order
    as the one-call backward, so the gradients are the same numbers.  Every parameter must sit in exactly one phase."""
    taps = [t for t, _ in phases if t is not None]
    head = [p for t, ps in phases if t is None for p in ps]
    grads = list(torch.autograd.grad(loss, taps + head, allow_unused=True))
    tap_grads, head_grads = grads[:len(taps)], grads[len(taps):]
    for p, g in zip(head, head_grads):
        if g is not None:
            p.grad = g if p.grad is None else p.grad + g
    k = 0
    for i, (tap, params) in enumerate(phases):
        if tap is not None:
            g, k = tap_grads[k], k + 1
            if g is not None:
                torch.autograd.backward([tap], [g], inputs=list(params))
        if after_phase is not None:
            after_phase(i)
        if between is not None and i + 1 < len(phases):
            between(i)


def _check_phase_buckets(fp, phases):
    """Phase i of the backward pass must complete exactly bucket i of the optimizer's flat layout."""
    for i, (_, params) in enumerate(phases):
        if i >= len(fp.bucket_params) or {id(fp.params[j]) for j in fp.bucket_params[i]} != {id(p) for p in params}:
            raise RuntimeError("the optimizer's gradient buckets do not match the model's backward phases "
                               "(build it with grad_buckets=model.grad_buckets())")


def train_step(model, batch, optimizer, opt, comm_stream=None):
    """zero_grad / forward / backward / [gradient all-reduce] / [global clip] / step (train.py:141-151).
    Returns (loss, loss_dict).  `optimizer` needs zero_grad(), step() and - for the data-parallel branch - `.fp`
    (optimization.FlatParams): nothing here is GPU-specific, the CPU tests drive it with a toy model over gloo.
    Data parallel with a model that exposes forward_phased and an optimizer laid out in gradient buckets: the backward pass
    runs tower by tower and each tower's all-reduce is issued as it completes (dist.BucketedGradSync), on comm_stream when
    given (so that it runs beside the next tower's backward kernels); otherwise one all-reduce of the whole flat buffer, enqueued
    on the current stream behind the backward pass."""
    optimizer.zero_grad()
    ddp = dist_info()[1] >= DDP_MIN_WORLD
    fp = getattr(optimizer, "fp", None)
    if ddp and hasattr(model, "forward_phased") and len(fp.bucket_ranges) > 1:
        from . import dist as ddist
        loss, loss_dict, phases = model.forward_phased(batch)
        _check_phase_buckets(fp, phases)
        sync = ddist.BucketedGradSync(fp, comm_stream=comm_stream)
        backward_in_phases(loss, phases, after_phase=sync.bucket_ready)
        sync.finish()
        if "loss_overall" not in loss_dict:
            ddist._c(None).host_wait(what="training step (bucketed all-reduce) before float(loss)")
            loss_dict = {"loss_overall": float(loss.detach()), **loss_dict}
    else:
        loss, loss_dict = model(batch)
        loss.backward()
        if ddp:
            from . import dist as ddist
            ddist.sync_gradients(fp)
    if getattr(opt, "grad_clip", -1) != -1:
        torch.nn.utils.clip_grad_norm_(model.parameters(), opt.grad_clip)
    optimizer.step()
    _end_zero_arena()
    if ddp:
        # the eager step reads float(loss) inside the forward pass, i.e. behind the PREVIOUS step's all-reduce: wait for this
        # step's collective here, against the deadline (comm.host_wait), so that the next blocking read cannot hang on a dead peer
        from . import dist as ddist
        ddist._c(None).host_wait(what="eager data-parallel training step")
    return loss, loss_dict


def _end_zero_arena():
    """The step's zero arena (functional.begin_zero_arena, opened by DLDKD.forward_tensors) is closed with the step: later
    autograd calls outside a training step must not carve from it."""
    import sys
    f = sys.modules.get(__package__ + ".functional")
    if f is not None:
        f.end_zero_arena()


class _CapturedStep:
    """One hipGraph of the training step for one batch signature, with its static device inputs."""


class GraphedTrainStep:
    """train_step (method/train.py:141-151: zero_grad / forward / backward / optimizer step) replayed from a hipGraph.

    The eager step of this path is launch-bound: ~350 kernel launches for ~7 ms of kernel time at the TVR batch (128 videos /
    640 queries), so its wall time follows the host - 8 to 15 ms box to box.  Captured once per batch signature, the step
    costs one graph launch.  What had to leave the captured region, and where it went:
      * host scalars of a step (learning rates, the batch's labels, the reference's CPU torch.randint draws for the triplet
        negatives - made here in the reference's order - and the Philox (seed, offset) of the dropout masks) are written into
        ONE pinned staging slot (staging.PinnedRing) and reach ONE device buffer by one asynchronous copy enqueued just before
        the graph; the captured kernels read them from that buffer;
      * dropout calls bake only their offset inside the step (functional.PhiloxStepState), so replays draw fresh masks and
        a captured run reproduces the eager run with the same torch.manual_seed;
      * float(loss) - the reference's one sync per step - happens after the replay (or never: defer_loss_float).
    The graph key holds everything that is baked in: tensor shapes and the hard-negative mode - not the labels, and (round 6) not
    the epoch schedule's alpha / belta / KD weight: they are staged per step like the other host values (functional.ScheduleWords).
    Real loaders pad every batch to ITS longest caption / video (data.collate_train, as the reference does), so raw shapes change
    from batch to batch; the stepper therefore pads the word axis up to a multiple of 8, the clip axis up to a multiple of 32 (the
    masks already make padding exact) and - once a second distinct query count has been seen: Charades / ActivityNet, whose videos
    carry different numbers of captions - the QUERY axis up to a multiple of 32 (padding queries of one zero word; the fused losses
    stop at the real count), which leaves a handful of signatures.  A new key runs eagerly the first time and is captured the
    second time; at most `max_graphs` graphs are kept (least recently used is dropped), an evicted key is never captured again
    and after `max_captures` captures every unseen key stays eager.  Under train_epoch a key is captured twice, each capture with
    its own input buffers, and the next batch is staged into the idle set while the step runs (iterate, double_buffer).
    Data parallel (world >= 2), default: the same graphs as on one GPU, with ONE all-reduce of the flat gradient buffer enqueued on
    the step's stream between the backward graphs and the optimizer graph (comm.RcclComm: a plain enqueue, no watchdog thread, so
    it may sit between replays; the optimizer graph then computes the clip's norms from the reduced gradients).  Every form of
    the step - eager, single graph, tower graphs - issues the same collectives in the same order, so ranks whose batch signatures
    differ (one replays while another captures) still pair up.  With the optimizer laid out in gradient buckets (opt.ddp_bucketed_overlap,
    make_optimizer): the step is a chain of graphs, one per tower of the backward pass; each tower's all-reduce is issued from the
    comm stream as its segment has been launched and runs under the next segment; the optimizer update is the chain's last graph
    (_capture_segments)."""

    QUERY_STREAMS_HIGH_PRIORITY = os.environ.get("DLDKD_QUERY_PRIO", "0") == "1"
    EARLY_VIDEO_START = True        # replay: the video towers start behind the video features' copy, not behind the whole batch's
    BWD_ORDER_REVERSED = os.environ.get("DLDKD_BWD_REV", "0") == "1"
    LOSS_SUM_IN_OPT_GRAPH = True
    GATHER_WITH_NORMS = True        # one GPU: a tower's gather also accumulates the clip's sums of squares (see _capture_parallel)
    UNIT_BRANCH_GRADS = True        # the branch graphs pass the constant 1 as the terms' upstream gradient (see branch_runner)
    TENSOR_KEYS = ("student_videos", "student_videos_mask", "teacher_videos", "student_text", "student_text_mask", "teacher_text")

    WATCH_EVERY = 32       # deferred-loss data-parallel replays between two progress markers (comm.watch)
    # The scalars the epoch schedule moves (alpha, belta, the KD weight: train.py:66-113) as device words the captured loss launches
    # read by address (functional.ScheduleWords), rewritten before the first replay of an epoch: ONE capture serves every epoch.
    # False: the values are part of the graph's key (a capture per epoch, eager steps after max_captures epochs).
    SCHEDULE_WORDS = True

    def __init__(self, model, optimizer, opt, max_graphs=8, defer_loss_float=False, max_captures=24):
        self.model, self.optimizer, self.opt = model, optimizer, opt
        self.max_graphs, self.defer, self.max_captures = max_graphs, defer_loss_float, max_captures
        self.graphs, self.seen, self.evicted = {}, {}, set()
        self.replays = self.eager_steps = self.captures = 0
        # Double-buffered inputs (iterate() / __call__(batch, next_batch=...)): a batch signature is captured TWICE, each capture
        # with its own static input buffers, and steps alternate between the two.  While step i runs from one set, batch i + 1 is
        # gathered / uploaded and copied into the other on `stage_stream` - the 295 MB of a TVR batch (0.10 ms of copy, 0.10 ms of
        # gather) leave the step's critical path.  One buffer set cannot do that: the input projection's backward pass, last kernel
        # of a tower's chain, still reads the raw features (columns with a small LayerNorm gamma are recomputed from x).
        self.double_buffer = False
        self.stage_stream = None
        self._turn, self._pending, self._lookahead, self._fetched = {}, None, None, None
        self.prefetched = 0                  # steps whose inputs were in place when the step began
        self.vary_queries, self._nq_seen = False, set()      # see _bucket_queries
        # Self-check of every newly captured graph (opt.graph_self_check, default on): the capture step runs the batch TWICE from the
        # same optimizer / RNG state - eagerly and as the first replay - and keeps the graph only if loss and parameters agree.  The
        # multi-graph stepper leans on hipGraph behaviour that has changed between ROCm point releases (a MEMSET node that left stale
        # words, a launch that faulted with three forks): a disagreement or a failed capture drops to the single-graph stepper, then
        # to eager steps, with a log line - never silently wrong, never an abort.
        # (one process only: under data parallelism the check would add a second gradient all-reduce to the capture step of ONE rank -
        # batch signatures differ between ranks - and the ranks' collectives would no longer pair up)
        self.self_check = bool(getattr(opt, "graph_self_check", True)) and dist_info()[1] < 2
        self.check_failures, self.capture_failures, self.fallbacks, self.last_check = 0, 0, [], None
        # EVERY step of this object - eager, capture, replay - runs on this side stream.  Autograd binds a parameter's
        # gradient-accumulation node to the stream that was current when the node was created and keeps it for as long as
        # any graph that reaches it is alive; a node born on the default stream (an eager step whose loss tensors the caller
        # still holds) would pull the default stream into a later capture on another stream through the engine's
        # cross-stream event, and hipStreamEndCapture then dies on the unjoined stream (seen as a segfault).
        self.stream = None
        self.comm_stream = None      # the bucketed overlap's collectives (dist.BucketedGradSync); the single all-reduce needs none
        # the four towers on four streams (model._encode_towers): fork / join edges of the captured graph, so the chip runs
        # their under-filled kernels side by side (C3 bf16 step 5.9 -> 5.3 ms); opt.tower_streams = False keeps one stream
        if hasattr(model, "tower_streams"):
            model.tower_streams = bool(getattr(opt, "tower_streams", True))
        # one GPU: the towers as separate graphs replayed on their own streams (_capture_parallel); opt.parallel_tower_graphs =
        # False keeps the single graph with fork / join edges
        self.parallel_towers = bool(getattr(opt, "parallel_tower_graphs", True))

    # -- what is baked into a graph
    def _key(self, batch):
        shapes = tuple((k, tuple(batch[k].shape), str(batch[k].dtype)) for k in self.TENSOR_KEYS)
        return (shapes, int(batch["student_text"].shape[0])) + self._key_tail(batch["student_videos"].is_cuda)

    def _key_tail(self, on_gpu):
        """The part of a graph's key that is not the batch's shapes: the model / run state baked into the captured launches."""
        m = self.model
        cfg = m.config
        get = (lambda k: cfg.get(k)) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k, None))
        sched = (None, None, None) if self._schedule_on_device(on_gpu) else (float(m.alpha), float(m.belta), float(m.weight))
        return (*sched, bool(get("use_hard_negative")), get("hard_pool_size"),
                m.label_style, bool(m.training), dist_info()[1] >= DDP_MIN_WORLD, getattr(self.opt, "grad_clip", -1),
                len(self.optimizer.fp.bucket_ranges))

    def _schedule_on_device(self, on_gpu):
        """Whether the step's losses run as the fused branch launches that can read the schedule from device words (model.py's
        condition for F_.branch_losses)."""
        from . import functional as F_
        return bool(self.SCHEDULE_WORDS and F_.BRANCH_LOSS_FUSED and F_.simpool_train_ok() and on_gpu and hasattr(self.model, "kl_intra_weight"))

    def _bucketed(self, batch):
        """The batch with its word axis padded to a multiple of 8 (at most max_desc_l) and its clip axis to a multiple of 32 (at
        most max_ctx_l): zero features, zero mask.  Padded positions are masked out of attention, pooling, max-pool and KL
        exactly (exp(-10000) and exp(-1e10) are 0 in fp32), so the step computes what it computes on the unpadded batch."""
        import torch.nn.functional as TF
        cfg = self.model.config
        get = (lambda k, d: cfg.get(k, d)) if isinstance(cfg, dict) else (lambda k, d: getattr(cfg, k, d))
        lq, lv = batch["student_text"].shape[1], batch["student_videos"].shape[1]
        lq_b, lv_b = self._bucket_lens(lq, lv)
        nq = int(batch["student_text"].shape[0])
        nq_b = self._bucket_queries(nq, batch["student_text"].is_cuda)
        if lq_b == lq and lv_b == lv and nq_b == nq:
            return batch
        out = dict(batch)
        if nq_b != nq:
            # padding queries behind the real ones: zero features, one valid (zero) word, a zero teacher vector; text_labels keeps
            # the real list - the model runs its losses over the first len(text_labels) rows (DLDKD.forward_tensors)
            for k in ("student_text", "teacher_text"):
                out[k] = TF.pad(batch[k], (0, 0, 0, 0, 0, nq_b - nq))
            out["student_text_mask"] = TF.pad(batch["student_text_mask"], (0, 0, 0, nq_b - nq))
            out["student_text_mask"][nq:, 0] = 1.0
            batch = out
            out = dict(batch)
        if lq_b != lq:
            out["student_text"] = TF.pad(batch["student_text"], (0, 0, 0, lq_b - lq))
            out["student_text_mask"] = TF.pad(batch["student_text_mask"], (0, lq_b - lq))
        if lv_b != lv:
            for k in ("student_videos", "teacher_videos"):
                out[k] = TF.pad(batch[k], (0, 0, 0, lv_b - lv))
            out["student_videos_mask"] = TF.pad(batch["student_videos_mask"], (0, lv_b - lv))
        return out

    def _streams(self, dev):
        if self.stage_stream is None:
            self.stage_stream = torch.cuda.Stream(device=dev)
        return self.stage_stream

    def enable_double_buffer(self):
        if not self.double_buffer:
            self.double_buffer = True
            self.max_graphs, self.max_captures = 2 * self.max_graphs, 2 * self.max_captures

    def iterate(self, loader):
        """The loader's batches, fetched ONE AHEAD on the staging stream (a device-resident loader's gather kernels, a host
        loader's uploads): `for batch in stepper.iterate(loader): stepper(batch)` - the step on batch i then finds batch i + 1
        and copies it into the other input buffer set while step i runs (double_buffer)."""
        dev = torch.device(self.opt.device)
        if dev.type != "cuda":
            yield from loader
            return
        self.enable_double_buffer()
        st = self._streams(dev)
        if hasattr(loader, "plans") and hasattr(getattr(loader, "devset", None), "gather"):
            # a device-resident training set (data.DeviceTrainLoader): the batch's shapes are known from its host-side plan, so its
            # rows are gathered STRAIGHT into the input buffers of the capture that will replay it (no staging copy at all), on the
            # staging stream, when the generator is resumed - i.e. behind the launch of the step before it, beside its kernels
            try:
                for plan in loader.plans():
                    yield self._materialize(loader.devset, plan, st)
            finally:
                self._fetched = None
            return
        it = iter(loader)

        def fetch():
            with torch.cuda.stream(st):
                b = next(it, None)
                if b is None:
                    return None
                b = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in b.items()}
                ev = torch.cuda.Event()
                ev.record(st)
            return b, ev

        nxt = fetch()
        try:
            while nxt is not None:
                cur = nxt
                nxt = fetch()
                self._lookahead = None if nxt is None else (nxt[0], st)
                self._fetched = (self._sig(cur[0]), cur[1])
                yield cur[0]
        finally:
            self._lookahead = self._fetched = None

    # the query axis is padded to a multiple of this (0: never) where the fused losses can skip the padding rows
    QUERY_BUCKET = int(os.environ.get("DLDKD_QUERY_BUCKET", "32"))

    def _bucket_queries(self, nq, on_gpu):
        """Batches of variable caption counts (data_provider.py:34-72: Charades ~2.3, ActivityNet ~3.7 captions per video) have a
        different number of queries each; padded to a bucket, a handful of captures serve them all (instead of eager steps).
        Padding starts with the SECOND distinct query count this stepper sees (vary_queries, sticky): a run whose batches all hold
        the same number of queries (TVR: 5 per video) never pays for padding rows (C5's 257-query bench batch: +7 % at 288 rows)."""
        b = int(self.QUERY_BUCKET)
        if b <= 0 or not self._schedule_on_device(on_gpu):
            return nq
        if not self.vary_queries:
            self._nq_seen.add(int(nq))
            self.vary_queries = len(self._nq_seen) > 1
        return -(-nq // b) * b if self.vary_queries else nq

    def _bucket_lens(self, lq, lv):
        cfg = self.model.config
        get = (lambda k, d: cfg.get(k, d)) if isinstance(cfg, dict) else (lambda k, d: getattr(cfg, k, d))
        return (max(min(-(-lq // 8) * 8, int(get("max_desc_l", lq))), lq), max(min(-(-lv // 32) * 32, int(get("max_ctx_l", lv))), lv))

    def _materialize(self, devset, plan, st):
        lq_b, lv_b = self._bucket_lens(plan.lmax["student_text"], max(plan.lmax["student_videos"], plan.lmax["teacher_videos"]))
        pad = {"student_videos": lv_b, "teacher_videos": lv_b, "student_text": lq_b}
        nv, f32 = plan.n["student_videos"], str(torch.float32)
        nq = self._bucket_queries(plan.n["student_text"], True)
        shape = {"student_videos": (nv, lv_b, plan.dim["student_videos"]), "student_videos_mask": (nv, lv_b),
                 "teacher_videos": (nv, lv_b, plan.dim["teacher_videos"]), "student_text": (nq, lq_b, plan.dim["student_text"]),
                 "student_text_mask": (nq, lq_b), "teacher_text": (nq, plan.lmax["teacher_text"], plan.dim["teacher_text"])}
        key = (tuple((k, shape[k], f32) for k in self.TENSOR_KEYS), nq) + self._key_tail(True)
        key = key + (self._turn.get(key, 0),)
        e = self.graphs.get(key) if getattr(self.opt, "grad_clip", -1) == -1 else None
        with torch.cuda.stream(st):
            if e is not None:
                st.wait_event(e.ev_done)                  # that capture's previous replay (two steps back) has read its inputs
            batch = devset.gather(plan, out=None if e is None else e.static, pad=pad, n_queries=nq)
            ev = torch.cuda.Event()
            ev.record(st)
        sig = self._sig(batch)
        if e is not None:
            e.ev_staged.record(st)
            e.staged_once = True
            self._pending = (sig, key, e, batch)
        self._fetched = (sig, ev, e)
        return batch

    def _sig(self, batch):
        return tuple(batch[k].data_ptr() for k in self.TENSOR_KEYS) + (id(batch["text_labels"]),)

    def __call__(self, batch, next_batch=None):
        """One training step.  next_batch: the batch of the NEXT call, if the caller has it already (device tensors, or host tensors to
        upload): its inputs are staged into the other buffer set while this step runs (enables double buffering)."""
        dev = batch["student_videos"].device
        if self.stream is None:
            if self.parallel_towers and getattr(self.model, "tower_streams", False) and hasattr(self.model, "_side_streams"):
                # this stepper's stream and the model's three side streams on four DIFFERENT hardware queues (measured once)
                from .staging import concurrent_streams
                # (QUERY_TOWERS_FIRST: towers 0, 1 - this stream and the first side stream - are the query towers)
                st = concurrent_streams(dev, 4, high_priority=2 if self.QUERY_STREAMS_HIGH_PRIORITY else 0)
                self.stream, self.model._side_streams = st[0], list(st[1:4])
            else:
                self.stream = torch.cuda.Stream(device=dev)
            self.comm_stream = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        self.stream.wait_stream(cur)
        look = self._lookahead
        if next_batch is not None:
            self.enable_double_buffer()
            look = (next_batch, cur)
        fetched = getattr(self, "_fetched", None)
        mine = fetched is not None and fetched[0] == self._sig(batch)
        if mine:
            # a batch iterate() fetched on the staging stream: if this step reads its tensors itself (no prefetch: eager, capture,
            # first replays) it has to wait for them, and their memory must outlive this stream's use of it
            self.stream.wait_event(fetched[1])
            for k in self.TENSOR_KEYS:
                batch[k].record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            out = self._step(batch, look)
            if mine and len(fetched) > 2 and fetched[2] is not None:
                # the batch's tensors ARE the input buffers of a capture (gathered straight into them): whatever form this step took
                # - that capture's replay, or an eager step reading them as plain tensors - the staging stream may refill them
                # only behind it
                fetched[2].ev_done.record(self.stream)
        cur.wait_stream(self.stream)
        return out

    def _eager(self, batch):
        self.eager_steps += 1
        loss, d = train_step(self.model, batch, self.optimizer, self.opt, comm_stream=self.comm_stream)
        # detached, like the replayed steps' outputs: the backward pass is over, nothing should keep the tape alive
        return loss.detach(), {k: (v.detach() if torch.is_tensor(v) else v) for k, v in d.items()}

    def _step(self, batch, look=None):
        if getattr(self.opt, "grad_clip", -1) != -1:
            return self._eager(batch)                     # clip_grad_norm_ syncs: not capturable
        pend, self._pending = self._pending, None
        if pend is not None and pend[0] == self._sig(batch) and pend[1][2:-1] == self._key_tail(True) and pend[1] in self.graphs:
            # this batch's inputs are already in the buffers of entry pend[2] (copied there while the previous step ran; the model's
            # baked-in state has not moved since)
            key, e = pend[1], pend[2]
            self.graphs[key] = self.graphs.pop(key)
            self._turn[key[:-1]] = 1 - key[-1]
            self.prefetched += 1
            out = self._replay(e, batch, staged=True)
        else:
            out = self._step_unstaged(batch)
        if look is not None and self.double_buffer:
            self._prefetch(*look)
        return out

    def _step_unstaged(self, batch):
        batch = self._bucketed(batch)
        key = self._key(batch)
        if self.double_buffer:                            # the signature's two captures take turns (see __init__)
            slot = self._turn.get(key, 0)
            self._turn[key] = 1 - slot
            key = key + (slot,)
        e = self.graphs.get(key)
        if e is None:
            if key in self.evicted or self.captures >= self.max_captures:
                return self._eager(batch)                 # capturing costs far more than an eager step: never twice for one key
            if len(self.seen) > 4096:                     # data whose every batch has its own signature: stay eager, stay small
                self.seen.clear()
            self.seen[key] = self.seen.get(key, 0) + 1
            if self.seen[key] < 2:                        # first sight: eager (also loads every kernel the graph needs)
                return self._eager(batch)
            return self._capture_checked(batch, key)
        else:
            self.graphs[key] = self.graphs.pop(key)       # most recently used last
        return self._replay(e, batch)

    def _prefetch(self, nb, src_stream):
        """Copy the NEXT step's inputs into the buffers of the entry that step will replay, on the staging stream: behind that
        entry's previous replay (two steps back) and the producer of the tensors, beside the step that has just been launched."""
        dev = torch.device(self.opt.device) if not torch.is_tensor(nb.get("student_videos")) or not nb["student_videos"].is_cuda \
            else nb["student_videos"].device
        st = self._streams(dev)
        if src_stream is not st:
            st.wait_stream(src_stream)
        sig = self._sig(nb)
        with torch.cuda.stream(st):
            b = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in nb.items()}
            b = self._bucketed(b)
            key = self._key(b)
            key = key + (self._turn.get(key, 0),)
            e = self.graphs.get(key)
            if e is None:
                return                                     # not captured (yet): that step stages its inputs itself
            st.wait_event(e.ev_done)
            for k in self.TENSOR_KEYS:
                e.static[k].copy_(b[k], non_blocking=True)
            e.ev_staged.record(st)
            e.staged_once = True
        self._pending = (sig, key, e, nb)               # (the reference keeps the tensors' addresses from being reused meanwhile)

    # -- capture with a guarded first replay
    def _snapshot(self):
        o = self.optimizer
        dev = o.fp.flat.device
        gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
        return {"flat": o.fp.flat.clone(), "m": o.m.clone(), "v": o.v.clone(), "step": o.step_count, "cpu_rng": torch.get_rng_state(),
                "gen_off": gen.get_offset(), "gen": gen}

    def _restore(self, st, params_too=True):
        from . import ops
        o = self.optimizer
        if params_too:
            o.fp.flat.copy_(st["flat"]); o.m.copy_(st["m"]); o.v.copy_(st["v"])
            ops.bump_param_epoch()
        o.step_count = st["step"]
        torch.set_rng_state(st["cpu_rng"])
        st["gen"].set_offset(st["gen_off"])

    def _capture_checked(self, batch, key):
        """Capture `key`; with the self-check on, the batch is stepped eagerly first (the reference), the state is rewound, the graph
        is captured and replayed once from the same state, and loss + parameters are compared.  Returns the step's result."""
        if not self.self_check:
            # Only the CAPTURE is guarded: it executes nothing, so the eager step that replaces it is this batch's first.  A failure
            # inside the replay - graphs already launched, perhaps the optimizer's, and under data parallelism this rank's
            # all-reduce already issued - must not be followed by a second step on the same batch: it propagates.
            try:
                e = self._capture(batch, key)
            except Exception as ex:   # noqa: BLE001
                return self._capture_failed(batch, key, ex, None)
            return self._replay(e, batch)
        st0 = self._snapshot()
        ref_loss, ref_dict = self._eager(batch)
        self.eager_steps -= 1                              # (bookkeeping: this eager run is the check's reference, not a step of its own)
        ref = {"flat": self.optimizer.fp.flat.clone(), "m": self.optimizer.m.clone(), "v": self.optimizer.v.clone(),
               "step": self.optimizer.step_count, "cpu_rng": torch.get_rng_state(), "gen_off": st0["gen"].get_offset(), "gen": st0["gen"]}
        ref_grad = self.optimizer.fp.grad.clone()
        self._restore(st0)
        try:
            e = self._capture(batch, key)
            loss, d = self._replay(e, batch)
        except Exception as ex:   # noqa: BLE001
            return self._capture_failed(batch, key, ex, (ref, ref_loss, ref_dict))
        o = self.optimizer
        lr = max(o.get_lr() + [0.0]) if hasattr(o, "get_lr") else 0.0
        dl = abs(float(loss) - float(ref_loss))
        # The GRADIENTS are compared tensor by tensor (relative l2 against the eager step's), the parameters by their mean move:
        # BertAdam has no bias correction (optimization.py:278-343), so in the first steps of a run an element's update is
        # ~3 lr sign(g) whatever |g| is - an element whose gradient is rounding noise around zero (the fp32 atomics' order differs
        # from run to run) flips sign between two correct steps and moves by 6 lr: max |d param| says nothing there (it rejected
        # correct captures at step 2 of TVR-shaped bf16 runs), a wrong gradient tensor shows up in its own norm
        dg_worst, dp_mean = self._grad_disagreement(ref_grad), float((o.fp.flat - ref["flat"]).abs().mean())
        lr_ref = max(lr, float(o._base_lr[0]) if getattr(o, "_base_lr", None) else 0.0)
        finite = bool(torch.isfinite(o.fp.flat).all())
        ok = finite and dl <= 5e-3 * max(abs(float(ref_loss)), 1e-3) and dg_worst <= self.CHECK_GRAD_TOL and dp_mean <= 1e-6 + 0.05 * lr_ref
        self.last_check = {"d_loss": dl, "worst_grad_rel_l2": dg_worst, "mean_d_param": dp_mean, "finite": finite}
        if ok:
            return loss, d
        self.check_failures += 1
        self.graphs.pop(key, None)
        self.captures -= 1
        self._restore(ref)                                 # the eager step's result stands
        self._degrade(key, f"replay disagrees with the eager step (|d loss| = {dl:.3e}, worst gradient tensor rel. l2 = {dg_worst:.3e}, "
                           f"mean |d param| = {dp_mean:.3e}, finite = {finite})")
        return ref_loss, ref_dict

    CHECK_GRAD_TOL = 5e-3      # a captured step's gradient tensors against the eager step's, relative l2 (measured: 1e-6 .. 1e-4)

    def _grad_disagreement(self, ref_grad):
        """max over the parameter tensors of ||g - g_ref|| / (||g_ref|| + 1e-6 max_t ||g_ref_t||) of the flat gradient buffer."""
        fp = self.optimizer.fp
        g = fp.grad
        if not bool(torch.isfinite(g).all()):
            return float("inf")
        zero = torch.zeros(1, dtype=torch.float64, device=g.device)
        c_d = torch.cat([zero, torch.cumsum((g - ref_grad).double() ** 2, 0)])
        c_r = torch.cat([zero, torch.cumsum(ref_grad.double() ** 2, 0)])
        s = torch.as_tensor(fp._starts, dtype=torch.int64, device=g.device)
        n = torch.as_tensor(fp._numels, dtype=torch.int64, device=g.device)
        num = (c_d[s + n] - c_d[s]).clamp_min(0).sqrt()
        den = (c_r[s + n] - c_r[s]).clamp_min(0).sqrt()
        return float((num / (den + 1e-6 * den.max() + 1e-30)).max())

    def _degrade(self, key, why):
        """One notch down: parallel tower graphs -> single graph -> this key stays eager."""
        if self.parallel_towers:
            self.parallel_towers = False
            self.seen[key] = 1                             # captured again (as a single graph) at its next sight
            what = "falling back to the single-graph stepper"
        else:
            self.evicted.add(key)
            what = "this batch signature stays eager"
        self.fallbacks.append((why, what))
        logger.warning(f"GraphedTrainStep: {why}: {what}")

    def _capture_failed(self, batch, key, ex, ref):
        from . import functional as F_
        self.capture_failures += 1
        F_.set_philox_step(None)
        if hasattr(self.model, "_tower_runner"):
            self.model._tower_runner = self.model._branch_runner = None
        self.graphs.pop(key, None)
        self._degrade(key, f"capture failed ({type(ex).__name__}: {str(ex)[:200]})")
        if ref is not None:
            st, loss, d = ref
            self._restore(st)
            return loss, d
        return self._eager(batch)

    # -- staging layout (int32 words): [philox 4][lr n_t][labels nq][per triplet call: r_t2v nq, r_v2t nv][schedule words: cq nq, cv nv, 10]
    def _layout(self, e, nq, nv, n_calls, sched=False):
        from . import functional as F_
        n_t = len(self.optimizer.fp.params)
        e.off = {"philox": 0, "lr": 4, "labels": 4 + n_t}
        o = 4 + n_t + nq
        for c in range(n_calls):
            e.off[("t2v", c)], e.off[("v2t", c)] = o, o + nq
            o += nq + nv
        if sched:
            e.off["sched"] = o
            o += F_.ScheduleWords.words_needed(nq, nv)
        e.words = o

    def _capture(self, batch, key):
        from . import functional as F_
        from .staging import PinnedRing, memset_node_defect
        m, opt_ = self.model, self.optimizer
        dev = batch["student_videos"].device
        # the hipGraph memset-node defect is probed on THIS runtime (once per process; never while a capture is open) and logged;
        # opt.scratch_zeroing = "probe" lets the result choose how the captured optimizer zeroes its norm scratch (memset node if
        # clean), the default "kernel" keeps the kernel fill
        self.memset_defect = memset_node_defect(dev, log=logger.info, select=getattr(self.opt, "scratch_zeroing", "kernel") == "probe")
        e = _CapturedStep()
        labels = list(batch["text_labels"])
        nq, nv = int(batch["student_text"].shape[0]), batch["student_videos"].shape[0]      # (nq: text rows, >= len(labels) when padded)
        e.hard = bool(key[5])
        e.n_calls = 2 if m.double_branch else 1
        self._layout(e, nq, nv, e.n_calls, sched=key[2] is None)
        e.ring = PinnedRing(4 * e.words, dev)
        e.dev_words = torch.zeros(e.words, dtype=torch.int32, device=dev)
        view = lambda name, n: e.dev_words[e.off[name]:e.off[name] + n]     # noqa: E731
        e.labels_dev = view("labels", nq)
        e.draws = [(view(("t2v", c), nq), view(("v2t", c), nv)) for c in range(e.n_calls)]
        e.philox = F_.PhiloxStepState(view("philox", 4).view(torch.int64))
        e.static = {k: torch.empty_like(batch[k]) for k in self.TENSOR_KEYS}
        e.static["text_labels"] = labels                 # only len() of it is baked in; the values are staged per step
        e.nq, e.nv = nq, nv
        e.ddp = bool(key[9])
        e.ev_done, e.ev_staged, e.staged_once = torch.cuda.Event(), torch.cuda.Event(), False
        e.sched = None
        if key[2] is None:
            # the schedule's scalars live in the step's staged words: written on the host into the pinned slot, uploaded with it
            n_s = F_.ScheduleWords.words_needed(nq, nv)
            e.sched = F_.ScheduleWords(nq, nv, m.label_style == "soft", dev, store=e.dev_words[e.off["sched"]:e.off["sched"] + n_s])
            for f in (m.kl_intra_weight, 0.0):              # the two branches' KL factors (model.py:143-155)
                e.sched.words_for(f)
            e.sched_in_slot = {}                            # ring slot -> the values it holds
        elif len(labels) != nq:
            raise RuntimeError("GraphedTrainStep: a padded query axis without the schedule words")
        old_lr = opt_.t_lr
        opt_.t_lr = view("lr", len(opt_.fp.params)).view(torch.float32)      # the update kernel reads the staged rates
        e.t_lr = opt_.t_lr
        old = F_.set_philox_step(e.philox)
        sched_ctx = F_.schedule_words(e.sched)
        sched_ctx.__enter__()
        try:
            opt_.zero_grad()
            e.graph = torch.cuda.CUDAGraph()
            # thread_local: only THIS thread's unsafe calls fail while the capture is open (a loader thread may touch the runtime).
            # Kernels the autograd thread launches into the capturing stream are captured in either mode.
            # Data parallel: the all-reduce is ONE enqueue (comm.RcclComm) between the backward graphs and the optimizer graph
            # of whichever form the step takes; with gradient buckets the step is the chain of segments instead.
            if e.ddp and hasattr(m, "forward_phased") and len(opt_.fp.bucket_ranges) > 1:
                self._capture_segments(e)
            elif (self.parallel_towers and hasattr(m, "forward_phased") and hasattr(m, "_tower_runner")
                  and getattr(m, "tower_streams", False) and getattr(m, "double_branch", False)
                  and hasattr(opt_.fp, "gather_subset")):
                self._capture_parallel(e)
            else:
                with torch.cuda.graph(e.graph, stream=self.stream, capture_error_mode="thread_local"):
                    loss, parts = m.forward_tensors(e.static, staged=e)
                    loss.backward()
                    if e.ddp:
                        opt_.fp.rebind_grads()
                    else:
                        opt_.enqueue(upload_lr=False)
                    e.loss, e.parts = loss.detach(), {k: (v.detach() if torch.is_tensor(v) else v) for k, v in parts.items()}
                del loss, parts
        except Exception:
            opt_.t_lr = old_lr
            raise
        finally:
            sched_ctx.__exit__(None, None, None)
            F_.set_philox_step(old)
        if e.sched is not None and not e.sched.used:
            raise RuntimeError("GraphedTrainStep: the captured step did not read the schedule words (alpha / belta / KD weight would be "
                               "frozen at this epoch's values): set GraphedTrainStep.SCHEDULE_WORDS = False")
        while len(self.graphs) >= self.max_graphs:
            old_key = next(iter(self.graphs))
            self.graphs.pop(old_key)
            self.evicted.add(old_key)
        self.graphs[key] = e
        self.captures += 1
        return e

    def _capture_segments(self, e):
        """Data parallel: the step as a CHAIN of graphs over one memory pool - segment i ends where tower i's gradients are in
        their flat range, so the replay can start that bucket's all-reduce on the comm stream and launch segment i + 1 right
        behind it; the fused optimizer update is the last graph of the chain (it runs once every collective has been joined).
        Graphs of one pool replay in capture order, which is the only order _replay uses."""
        m, opt_ = self.model, self.optimizer
        e.segments, e.pool = [], torch.cuda.graph_pool_handle()
        ctx = [None]

        def open_segment():
            g = torch.cuda.CUDAGraph()
            c = torch.cuda.graph(g, pool=e.pool, stream=self.stream, capture_error_mode="thread_local")
            c.__enter__()
            ctx[0] = c
            e.segments.append(g)

        def close_segment(*exc):
            c, ctx[0] = ctx[0], None
            if c is not None:
                c.__exit__(*(exc or (None, None, None)))

        try:
            open_segment()
            loss, parts, phases = m.forward_phased(e.static, staged=e)
            _check_phase_buckets(opt_.fp, phases)
            backward_in_phases(loss, phases, after_phase=opt_.fp.rebind_bucket,
                               between=lambda i: (close_segment(), open_segment()))
            e.had = opt_.fp.rebind_grads()
            e.loss, e.parts = loss.detach(), {k: (v.detach() if torch.is_tensor(v) else v) for k, v in parts.items()}
            del loss, parts, phases
            close_segment()
            open_segment()
            opt_.enqueue(upload_lr=False)
            close_segment()
            e.opt_graph = e.segments.pop()
        except BaseException as ex:
            close_segment(type(ex), ex, ex.__traceback__)
            raise

    def _capture_parallel(self, e):
        """One GPU: the step as a handful of LINEAR graphs - [zero arena, lengths, teacher scores] -> T x [tower forward] ->
        2 x [one branch's losses + their backward pass down to its two tower outputs] -> T x [tower backward + gather of its
        gradients] -> [sum of the loss terms] -> [optimizer] - where every tower and every branch is captured on ITS OWN stream
        with its own memory pool and the replay launches them on those streams between events.  A single graph with fork / join edges leaves the
        overlap to the graph executor, which (ROCm 7.0.2) runs the four backward chains of the C3 step two at a time in a
        fixed order - the second video tower started when the first query tower was done (profiles/r03/
        step_timeline_bf16_graph.txt) - and crashes when three side streams fork from one point in several graphs of a
        process (DLDKD.TOWER_FORK).  Here every graph is a linear chain on one stream and the concurrency is that of T real
        streams.  Graphs of one pool are captured and replayed in the same order on one stream (forward i, backward i;
        the three main-stream graphs); tensors that cross pools (tower outputs, their gradients, saved activations, the zero
        arena) are alive while their consumer is captured, and nothing is captured into their pool afterwards that could run
        before that consumer."""
        from . import functional as F_
        m, opt_ = self.model, self.optimizer
        dev = e.static["student_videos"].device
        ctx = [None]
        pools = {}

        def pool(k):
            if k not in pools:
                pools[k] = torch.cuda.graph_pool_handle()
            return pools[k]

        def open_graph(stream, pool_key):
            g = torch.cuda.CUDAGraph()
            c = torch.cuda.graph(g, pool=pool(pool_key), stream=stream, capture_error_mode="thread_local")
            c.__enter__()
            ctx[0] = c
            return g

        def close_graph(*exc):
            c, ctx[0] = ctx[0], None
            if c is not None:
                c.__exit__(*(exc or (None, None, None)))

        e.par = {"fwd": [], "bwd": [], "streams": [], "loss": []}
        stream_of, tap_grads = {}, {}
        unit = F_.unit_grad(dev)

        def runner(thunks, weights):
            close_graph()                                   # the graph in front of the towers ends here
            # launch order of the replay: the towers with the most input first (the graphs of a phase reach the GPU one after
            # the other: the long video towers must not be last)
            e.par["order"] = sorted(range(len(thunks)), key=lambda i: -weights[i])
            e.par["video"] = [w == e.static["student_videos"].numel() and w != e.static["student_text"].numel() for w in weights]
            # tower i on the stream the eager step (model._encode_towers) runs it on: this stepper's stream and the model's
            # side streams - a parameter's gradient-accumulation node stays bound to the stream of its first use
            if m._side_streams is None or m._side_streams[0].device != dev:
                m._side_streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
            streams = ([self.stream] + list(m._side_streams))[:len(thunks)]
            if len(streams) < len(thunks):
                raise RuntimeError("parallel tower graphs: more towers than streams")
            outs = []
            for i, th in enumerate(thunks):
                e.par["fwd"].append(open_graph(streams[i], ("tower", i)))
                outs.append(th())
                close_graph()
                stream_of[id(outs[-1])] = i
            e.par["streams"] = streams
            return outs                                     # no graph is open: the model calls the branch runner next

        def branch_runner(parts):
            """parts = [(fn, (q, g))] per branch: fn(q, g) -> that branch's loss terms.  Each branch gets ONE graph - its losses
            and their backward pass down to the two tower outputs - on the stream of its query tower (branch 0: the main
            stream), reading the tower outputs through views made on that stream inside that graph, so that d loss / d view
            is captured at a node of the capturing stream (no cross-stream hand-over inside a capture)."""
            res = []
            for b, (fn, (q, g)) in enumerate(parts):
                iq, ig = stream_of[id(q)], stream_of[id(g)]
                st = e.par["streams"][iq]
                gr = open_graph(st, ("loss", b))
                qv, gv = q.view_as(q), g.view_as(g)
                terms = fn(qv, gv)
                if self.UNIT_BRANCH_GRADS and all(torch.is_tensor(t) and t.dim() == 0 and t.dtype == torch.float32 for t in terms):
                    # the step's loss is the plain sum of the terms (model.py:157-160): every term's upstream gradient is the
                    # constant 1 - handed over as such there is no add per term, no ones fill and (functional._BranchLoss) no
                    # scaling launch in the branch's graph
                    gq, gg = torch.autograd.grad(list(terms), [qv, gv], grad_outputs=[unit] * len(terms), allow_unused=True)
                    total = None
                else:
                    total = terms[0]
                    for t in terms[1:]:
                        total = total + t
                    gq, gg = torch.autograd.grad(total, [qv, gv], allow_unused=True)
                close_graph()
                e.par["loss"].append((gr, iq, (iq, ig)))
                tap_grads[iq], tap_grads[ig] = gq, gg
                res.append(terms)
                del total, qv, gv
            e.par["tail"] = open_graph(self.stream, "main")  # the sum of the two branches' terms
            return res

        import contextlib

        @contextlib.contextmanager
        def pre_ln_hook(dual):
            # dual: the model is about to enqueue the ONE launch that writes both video towers' LayerNorm rows.  It becomes a graph of
            # its own on the FIRST VIDEO TOWER's stream (and in that tower's pool: replayed there in front of the tower's forward
            # graph); the other video tower's stream waits for it.  On the main stream it would sit in front of the teacher scores
            # and a query tower - the chain the step's loss section waits for (measured: no gain there).
            if dual:
                if m._side_streams is None or m._side_streams[0].device != dev:
                    m._side_streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
                n_q = 2 if m.QUERY_TOWERS_FIRST else 0             # (double branch: towers [q, q, v, v] or [v, v, q, q])
                e.par["pre0_tower"] = n_q
                st = ([self.stream] + list(m._side_streams))[n_q]
                e.par["pre0"] = open_graph(st, ("tower", n_q))
                try:
                    yield
                finally:
                    close_graph()
            else:
                yield
            e.par["pre"] = open_graph(self.stream, "main")
            if e.par.get("zero_norms") is not None:
                e.par["zero_norms"]()

        m._tower_runner, m._branch_runner, m._pre_ln_hook = runner, branch_runner, pre_ln_hook
        try:
            # (the first main-stream graph is opened by pre_ln_hook, which the model enters first thing in forward_tensors)
            # the clip's per-tensor sums of squares are accumulated by the towers' gathers (one pass over a tower's gradients on that
            # tower's stream: FlatParams._gather) into the scratch zeroed here, at the head of the step; the optimizer graph is then
            # the update alone
            # (data parallel: the clip's norms are those of the all-reduced gradients - the optimizer graph computes them itself)
            want_norms = (self.GATHER_WITH_NORMS and not e.ddp and hasattr(opt_, "zero_norms")
                          and opt_.param_groups[0].get("max_grad_norm", -1) > 0)
            e.par["zero_norms"] = opt_.zero_norms if want_norms else None       # (enqueued by pre_ln_hook at the head of "pre")
            norm2 = opt_.norm2 if want_norms else None
            loss, parts, phases = m.forward_phased(e.static, staged=e)
            e.par.pop("zero_norms", None)
            if any(t is None or id(t) not in stream_of for t, _ in phases) or "tail" not in e.par:
                raise RuntimeError("parallel tower graphs: a backward phase is not one of the towers")
            e.loss, e.parts = loss.detach(), {k: (v.detach() if torch.is_tensor(v) else v) for k, v in parts.items()}
            close_graph()
            n = len(e.par["fwd"])
            e.par["bwd"], e.par["loss_of"] = [None] * n, [None] * n
            for b, (_, _, towers) in enumerate(e.par["loss"]):
                for t in towers:
                    e.par["loss_of"][t] = b
            for tap, params in phases:
                i = stream_of[id(tap)]
                e.par["bwd"][i] = open_graph(e.par["streams"][i], ("tower", i))
                if tap_grads.get(i) is not None:
                    torch.autograd.backward([tap], [tap_grads[i]], inputs=list(params))
                opt_.fp.gather_subset(params, norm2=norm2) if norm2 is not None else opt_.fp.gather_subset(params)
                close_graph()
            del loss, parts, phases, tap
            stream_of.clear()
            tap_grads.clear()
            e.par["opt"] = open_graph(self.stream, "main")
            # the step's loss value (the sum of the five terms, in the model's order) at the head of the optimizer graph instead of
            # in a graph of its own between the towers and the optimizer: one graph launch fewer on the serial tail
            keys = ("inher_trip", "inher_nce", "kl", "explore_trip", "explore_nce")
            if self.LOSS_SUM_IN_OPT_GRAPH and all(torch.is_tensor(e.parts.get(k)) and e.parts[k].is_cuda and e.parts[k].dim() == 0 for k in keys):
                with torch.no_grad():
                    e.loss = F_.sum_scalars(*[e.parts[k] for k in keys]).detach()
                e.par["tail_kept"], e.par["tail"] = e.par["tail"], None      # (kept alive: its pool holds the branch graphs' tensors)
            e.had = opt_.fp.rebind_grads()
            # opt.ddp_allreduce_in_graph (off by default): the gradient all-reduce as a NODE of the optimizer graph - RCCL enqueues
            # into the capturing stream like any launch (comm.RcclComm: tests/test_comm_gpu.py captures a gather) - instead of one
            # enqueue between the graphs: one launch boundary fewer on the serial tail.  Off by default because a captured collective
            # over more than one rank cannot be validated from a one-GPU environment.
            e.par["ar_in_graph"] = bool(e.ddp and getattr(self.opt, "ddp_allreduce_in_graph", False))
            if e.par["ar_in_graph"]:
                from . import dist as ddist
                ddist.all_reduce_flat(opt_.fp.grad)
            opt_.enqueue(upload_lr=False)
            close_graph()
            e.par["ev"] = {"fwd": [torch.cuda.Event() for _ in range(n)], "bwd": [torch.cuda.Event() for _ in range(n)],
                           "loss": [torch.cuda.Event() for _ in e.par["loss"]]}
            e.par["ev_pre"], e.par["ev_pre_done"], e.par["ev_in_video"] = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
            e.par["ev_ln"] = torch.cuda.Event()
        except BaseException as ex:
            close_graph(type(ex), ex, ex.__traceback__)
            raise
        finally:
            m._tower_runner = m._branch_runner = m._pre_ln_hook = None

    def _replay_parallel(self, e):
        par, main = e.par, self.stream
        streams, ev = par["streams"], par["ev"]
        # the towers read the staged batch only (the zero arena of the graph in front is for the backward passes, which wait for
        # a loss graph that runs behind it): the side streams start here, beside that graph's teacher scores
        par["ev_pre"].record(main)
        pre0 = par.get("pre0")
        if pre0 is not None:
            # both video towers' input LayerNorm rows from ONE launch (functional.in_proj_ln_dual), on the first video tower's stream
            # in front of that tower's forward graph; the other video tower waits for it
            s0 = streams[par["pre0_tower"]]
            s0.wait_event(par["ev_in_video"] if self.EARLY_VIDEO_START else par["ev_pre"])
            with torch.cuda.stream(s0):
                pre0.replay()
                par["ev_ln"].record(s0)
        par["pre"].replay()
        par["ev_pre_done"].record(main)
        for i in par["order"]:
            # (a tower on the main stream runs behind the graph in front anyway; a video tower on a side stream needs the video
            # features and the dropout state only.  Launching those two graphs BEFORE the host's other work for the step was tried:
            # C3 2.69 -> 2.63 ms but C5 1.42 -> 1.50 with float(loss) every step, no change without - the host's launch order is
            # then the wrong way round for the short C5 towers - not kept)
            early = self.EARLY_VIDEO_START and par["video"][i] and streams[i] is not main
            if streams[i] is not main:
                if pre0 is not None and par["video"][i]:
                    if i != par["pre0_tower"]:              # (the first video tower's graph follows pre0 on its own stream)
                        streams[i].wait_event(par["ev_ln"])
                else:
                    streams[i].wait_event(par["ev_in_video"] if early else par["ev_pre"])
            with torch.cuda.stream(streams[i]):
                par["fwd"][i].replay()
                ev["fwd"][i].record(streams[i])
        # (cross-stream waits only: an event recorded on the waiting stream itself is implied by the stream's order, and every wait
        # that is enqueued costs the stream ~7 us of barrier processing - the six waits in front of the optimizer graph were
        # 40 us of the step's serial tail: tools/graph_timeline.py)
        for b, (g, si, towers) in enumerate(par["loss"]):   # a branch's losses + their backward pass to its two tower outputs
            if streams[si] is not main:
                streams[si].wait_event(par["ev_pre_done"])  # lengths, teacher scores, zero arena
            for t in towers:
                if streams[t] is not streams[si]:
                    streams[si].wait_event(ev["fwd"][t])
            with torch.cuda.stream(streams[si]):
                g.replay()
                ev["loss"][b].record(streams[si])
        for i in (par["order"][::-1] if self.BWD_ORDER_REVERSED else par["order"]):
            if streams[par["loss"][par["loss_of"][i]][1]] is not streams[i]:
                streams[i].wait_event(ev["loss"][par["loss_of"][i]])
            with torch.cuda.stream(streams[i]):
                par["bwd"][i].replay()
                ev["bwd"][i].record(streams[i])
        for i, x in enumerate(ev["bwd"]):                   # (a branch's loss graph is followed by a tower's backward graph on its stream)
            if streams[i] is not main:
                main.wait_event(x)
        if par["tail"] is not None:
            par["tail"].replay()
        if e.ddp:
            # every tower's gradients are in the flat buffer (the waits above): the mean all-reduce is one enqueue on this stream
            # between the backward graphs and the optimizer graph (or a node of that graph: opt.ddp_allreduce_in_graph - the flag
            # check keeps its rank-invariant cadence either way)
            from . import dist as ddist
            if par.get("ar_in_graph"):
                ddist.check_had_flags(self.optimizer.fp, e.had)
            else:
                ddist.sync_gradients(self.optimizer.fp, had=e.had)
        par["opt"].replay()

    def _replay(self, e, batch, staged=False):
        from . import ops
        m, opt_ = self.model, self.optimizer
        if e.staged_once:
            self.stream.wait_event(e.ev_staged)           # the staging stream's last copy into these buffers (this batch's, if staged)

        def stage(keys):
            if staged:
                return
            for k in keys:                                # (a data path that fills e.static itself hands the same storage back: no copy)
                if batch[k].data_ptr() != e.static[k].data_ptr():
                    e.static[k].copy_(batch[k], non_blocking=True)

        n_t = len(batch["text_labels"])                   # the batch's real queries (<= e.nq text rows)
        if e.sched is None and n_t != e.nq:
            raise RuntimeError(f"GraphedTrainStep: {n_t} labels for a step captured with {e.nq} queries")
        # the video features first - the largest copy (201 MB at the TVR batch: 72 us) feeding the longest chains; the video towers
        # start behind it and the step's scalars (ev_in_video below) while the other inputs are still being copied
        stage(self.TENSOR_KEYS[:2])
        slot = e.ring.next()[:4 * e.words].view(torch.int32)
        e.philox.begin_step(slot[0:4].view(torch.int64))
        if e.sched is not None:
            # the epoch's alpha / belta / KD weight and the batch's query count -> this slot's schedule words (host writes; a slot
            # that already holds these values - every step of an epoch on a fixed-count dataset - is left alone)
            st = (float(m.alpha), float(m.belta), float(m.weight), n_t)
            if e.sched_in_slot.get(e.ring.i) != st:
                n_s = e.sched.words_needed(e.nq, e.nv)
                e.sched.write(slot[e.off["sched"]:e.off["sched"] + n_s], *st)
                e.sched_in_slot[e.ring.i] = st
        par = getattr(e, "par", None)
        if par:
            # ... and the dropout state (the slot's first 16 bytes) - all a tower's forward pass reads of the step's scalars - goes up
            # before the host draws the triplet negatives: a step that begins on an idle GPU (float(loss) every step) starts its
            # video towers ~0.1 ms earlier
            e.ring.upload_range(e.dev_words.view(torch.uint8), 0, 16, last=False)
            par["ev_in_video"].record(self.stream)
        opt_.t_lr = e.t_lr
        n_par = len(opt_.fp.params)
        opt_.host_prepare(lr_out=slot[e.off["lr"]:e.off["lr"] + n_par].view(torch.float32))
        labels_np = __import__("numpy").asarray(batch["text_labels"])      # THIS batch's caption -> video map
        slot[e.off["labels"]:e.off["labels"] + n_t] = torch.from_numpy(labels_np.astype("int32"))
        if n_t < e.nq:                                    # padding queries: label 0, draw 1 (never read by the losses)
            slot[e.off["labels"] + n_t:e.off["labels"] + e.nq] = 0
        for c in range(e.n_calls):                        # the reference's CPU draws, same order and arguments
            _, r_t2v, r_v2t = m._draw_triplet(labels_np, e.nv)
            slot[e.off[("t2v", c)]:e.off[("t2v", c)] + n_t] = r_t2v
            if n_t < e.nq:
                slot[e.off[("t2v", c)] + n_t:e.off[("t2v", c)] + e.nq] = 1
            if r_v2t is not None:
                slot[e.off[("v2t", c)]:e.off[("v2t", c)] + e.nv] = r_v2t
        if par:
            e.ring.upload_range(e.dev_words.view(torch.uint8), 16, 4 * e.words, last=True)
        else:
            e.ring.upload(e.dev_words.view(torch.uint8))
        stage(self.TENSOR_KEYS[2:])
        if par:
            opt_.fp.bind_views(e.had)                     # the captured copies fill the flat ranges: nothing to gather
            self._replay_parallel(e)
        elif getattr(e, "segments", None):
            from . import dist as ddist
            sync = ddist.BucketedGradSync(opt_.fp, comm_stream=self.comm_stream)
            opt_.fp.bind_views(e.had)                     # the captured copies fill the flat ranges: nothing to gather
            for i, g in enumerate(e.segments):
                g.replay()
                sync.issue(i)                             # tower i's all-reduce runs under segment i + 1
            sync.finish()
            e.opt_graph.replay()
        else:
            e.graph.replay()
            if e.ddp:
                from . import dist as ddist
                ddist.sync_gradients(opt_.fp)
                opt_.enqueue(upload_lr=False)
        if self.double_buffer:
            e.ev_done.record(self.stream)                 # the staging stream may overwrite these input buffers behind this point
        ops.bump_param_epoch()
        _end_zero_arena()
        self.replays += 1
        out = dict(e.parts)
        if e.ddp:
            # the host's blocking read of a data-parallel step is deadline-bounded (comm.host_wait: a dead peer parks the stream in
            # the all-reduce for ever); a loop that never reads the loss back leaves a progress marker every WATCH_EVERY replays
            from . import dist as ddist
            c = ddist._c(None)
            if not self.defer:
                c.host_wait(self.stream, what="replayed data-parallel training step")
            elif self.replays % self.WATCH_EVERY == 0:
                c.watch(self.stream, what=f"replayed data-parallel training steps (marker every {self.WATCH_EVERY})")
        loss_overall = e.loss if self.defer else float(e.loss)
        return e.loss, {"loss_overall": loss_overall, **out}


def make_train_loader(train_dataset, opt, rank=None, world=None):
    """Single process: the reference's shuffled loader (train.py:283-289).  Data parallel: a DistributedSampler cuts every
    epoch's permutation into disjoint per-rank parts (call loader.sampler.set_epoch(epoch) before each epoch: done by
    train()); bsz stays the per-rank batch of 128 videos, so steps per epoch - hence BertAdam's t_total - shrink by the
    world size, as they must for the warm-up/decay schedule to span the run."""
    if rank is None or world is None:
        rank, world = dist_info()
    sampler = None
    if world > 1:
        from torch.utils.data.distributed import DistributedSampler
        sampler = DistributedSampler(train_dataset, num_replicas=world, rank=rank, shuffle=True, seed=int(getattr(opt, "seed", 0) or 0),
                                     drop_last=False)
    # opt.device_resident_train: True / False / "auto" (the default): on a GPU, when the dataset hands out the same item every time
    # it is asked (the reference's Dataset4DLDKD does: features read and down-sampled deterministically, data_provider.py:34-72) and
    # the whole set fits opt.train_feature_cache_gb (160; TVR: 27 GB)
    mode = getattr(opt, "device_resident_train", "auto")
    if mode in (True, "auto") and torch.device(getattr(opt, "device", "cpu")).type == "cuda" and len(train_dataset) > 0:
        # the training set read once into ragged tables on the device, batches gathered there (data.DeviceTrainSet): same
        # batches in the same order with the same random draws, no per-epoch DataLoader / pad / H2D
        from .data import DeviceTrainLoader, DeviceTrainSet
        devset = None
        if mode == "auto" and not _items_repeat(train_dataset):
            logger.info("training set: items differ between two reads (augmentation?): the host DataLoader stays in charge")
        else:
            try:
                devset = DeviceTrainSet(train_dataset, opt.device, num_workers=opt.num_workers,
                                        cap_gb=float(getattr(opt, "train_feature_cache_gb", 160.0)))
            except (MemoryError, ValueError, TypeError, IndexError) as ex:
                if mode is True:
                    raise
                logger.info(f"training set not kept on the device ({type(ex).__name__}: {ex}): the host DataLoader stays in charge")
        if devset is not None:
            if sampler is not None:
                sampler = DistributedSampler(range(len(train_dataset)), num_replicas=world, rank=rank, shuffle=True,
                                             seed=int(getattr(opt, "seed", 0) or 0), drop_last=False)
            logger.info(f"training set device-resident: {devset.n_videos} videos, {devset.n_caps} captions, "
                        f"{sum(t.numel() for t in devset.src.values()) * 4 / 1e9:.2f} GB")
            return DeviceTrainLoader(devset, opt.bsz, shuffle=True, sampler=sampler)
    return DataLoader(train_dataset, batch_size=opt.bsz, shuffle=sampler is None, sampler=sampler, pin_memory=opt.pin_memory,
                      num_workers=opt.num_workers, collate_fn=collate_train)


def _items_repeat(dataset):
    """Two reads of the first and the last item give the same tensors (a dataset without random augmentation); the global RNG is
    left where it was."""
    import numpy as np
    rng = torch.get_rng_state()
    np_state = np.random.get_state()
    try:
        for i in sorted({0, len(dataset) - 1}):
            a, b = dataset[i], dataset[i]
            for x, y in zip(a[:4], b[:4]):
                xs, ys = (x, y) if isinstance(x, (list, tuple)) else ([x], [y])
                if len(xs) != len(ys) or any(not np.array_equal(np.asarray(u), np.asarray(v)) for u, v in zip(xs, ys)):
                    return False
        return True
    except Exception:   # noqa: BLE001 - a dataset this probe cannot read twice stays with the host loader
        return False
    finally:
        torch.set_rng_state(rng)
        np.random.set_state(np_state)


def seed_rank(opt, rank):
    """Rank-offset seed for what must differ between replicas: dropout masks (torch's CUDA generator, functional._philox_slot)
    and the triplet negatives (CPU torch.randint, model.py:366-380).  Parameters do not depend on it: they are broadcast."""
    seed = int(getattr(opt, "seed", 0) or 0) + rank
    torch.manual_seed(seed)
    return seed


def train_epoch(model, train_loader, optimizer, opt, epoch_i, training=True, stepper=None):
    """One epoch (train.py:52-183).  Returns the mean of every loss entry.  stepper: a GraphedTrainStep (or None: eager)."""
    from .data import host_threads
    with host_threads():
        return _train_epoch(model, train_loader, optimizer, opt, epoch_i, training, stepper)


def _train_epoch(model, train_loader, optimizer, opt, epoch_i, training=True, stepper=None):
    model.train(mode=training)
    if opt.hard_negative_start_epoch != -1 and epoch_i >= opt.hard_negative_start_epoch:
        model.set_hard_negative(True, opt.hard_pool_size)
    weight, alpha, belta = epoch_schedules(opt, epoch_i)
    if weight is not None:
        model.weight = weight
    if alpha is not None:
        model.alpha = alpha
    if belta is not None:
        model.belta = belta
    logger.info(f"Epoch {epoch_i}, Alpha: {model.alpha}, belta: {model.belta}")
    keys, acc, n = None, None, 0
    # a graphed stepper fetches one batch ahead on its staging stream and double-buffers the step's inputs (GraphedTrainStep.iterate)
    ahead = (training and stepper is not None and hasattr(stepper, "iterate") and getattr(opt, "prefetch_batches", True)
             and torch.device(opt.device).type == "cuda" and getattr(opt, "grad_clip", -1) == -1)
    for batch_idx, batch in enumerate(stepper.iterate(train_loader) if ahead else train_loader):
        batch = {k: (v.to(opt.device, non_blocking=True) if k != "text_labels" else v) for k, v in batch.items()}
        if training:
            _, loss_dict = (stepper or (lambda b: train_step(model, b, optimizer, opt)))(batch)
        else:
            with torch.no_grad():
                _, loss_dict = model(batch)
        # running sums stay on the device (one stack + one add per step): the reference reads 7 scalars back per step
        # (train.py:153-157), i.e. 7 host synchronisations; here the host reads once per epoch
        keys = keys or list(loss_dict)
        dev = torch.device(opt.device)
        vec = torch.stack([(loss_dict[k].detach().float() if torch.is_tensor(loss_dict[k])
                            else torch.tensor(float(loss_dict[k]), device=dev)).reshape(()).to(dev) for k in keys])
        acc = vec if acc is None else acc + vec
        n += 1
        if getattr(opt, "debug", False) and batch_idx == 3:
            break
    if acc is None:
        return {}
    if acc.is_cuda and dist_info()[1] > 1:
        from . import comm as _comm_mod
        _comm_mod.current().host_wait(what="end of a data-parallel training epoch (loss sums read-back)")
    return {k: float(v) / max(n, 1) for k, v in zip(keys, acc.cpu().tolist())}


def save_checkpoint(model, epoch_i, path):
    """{"model", "model_cfg", "epoch"} (train.py:231-235), loadable by the reference's setup_model and ours."""
    torch.save({"model": model.state_dict(), "model_cfg": model.config, "epoch": epoch_i}, path)


def load_checkpoint(path, opt, map_location=None):
    """Counterpart of setup_model (eval.py:266-283)."""
    from .model import DLDKD
    ck = torch.load(path, map_location=map_location, weights_only=False)
    model = DLDKD(ck["model_cfg"], opt)
    model.load_state_dict(ck["model"])
    return model, ck["epoch"]


def train(model, train_dataset, val_video_dataset, val_text_dataset, opt):
    """Epoch loop with eval after each epoch, best-checkpoint saving and early stop (train.py:191-247)."""
    rank, world = dist_info()
    model.to(opt.device)
    # opt.train_precision: "parity" / "fp32" (the default of the process), "mixed" (fp32-grade forward - the seven losses within 1e-6 of
    # the reference's - and the throughput mode's bf16 backward: 2x the parity step rate) or "bf16" (throughput); None: whatever
    # ops.set_gemm_precision was given.  Restored when train() returns.
    from . import ops as _ops
    tp = getattr(opt, "train_precision", None)
    prev_precision = _ops.precision_mode()
    if tp is not None:
        _ops.set_gemm_precision({"parity": "fp32"}.get(tp, tp))
        logger.info(f"training precision: {_ops.precision_mode()}")
    try:
        return _train(model, train_dataset, val_video_dataset, val_text_dataset, opt, rank, world)
    finally:
        _ops.set_gemm_precision(prev_precision)


def _train(model, train_dataset, val_video_dataset, val_text_dataset, opt, rank, world):
    loader = make_train_loader(train_dataset, opt, rank, world)
    optimizer = make_optimizer(model, opt, len(loader))
    if world > 1:
        from . import dist as ddist
        from .eval import eval_epoch_sharded
        ddist.broadcast_parameters(optimizer.fp)          # every replica starts from rank 0's weights
        seed_rank(opt, rank)
    best, es_cnt = 0.0, 0
    history = []
    # the step replayed from a hipGraph (GraphedTrainStep) unless opt.graph_step is False or the model is not on a GPU;
    # float(loss) per step is deferred: train_epoch reads the loss sums back once per epoch
    stepper = None
    if getattr(opt, "graph_step", True) and torch.device(opt.device).type == "cuda":
        stepper = GraphedTrainStep(model, optimizer, opt, defer_loss_float=True)
    for epoch_i in range(-1 if getattr(opt, "eval_untrained", False) else 0, opt.n_epoch):
        if loader.sampler is not None and hasattr(loader.sampler, "set_epoch"):
            loader.sampler.set_epoch(max(epoch_i, 0))
        losses = train_epoch(model, loader, optimizer, opt, epoch_i, training=True, stepper=stepper) if epoch_i > -1 else {}
        with torch.no_grad():
            # sharded eval: every rank encodes 1/world of the gallery; its SumR is the same number on every rank
            rsum = (eval_epoch_sharded if world > 1 else eval_epoch)(model, val_video_dataset, val_text_dataset, opt)
        history.append((epoch_i, losses, rsum))
        if rsum > best:
            best, es_cnt = rsum, 0
            if getattr(opt, "ckpt_filepath", None) and rank == 0:
                save_checkpoint(model, epoch_i, opt.ckpt_filepath)
        else:
            es_cnt += 1
            if opt.max_es_cnt != -1 and es_cnt > opt.max_es_cnt:
                break
        if getattr(opt, "debug", False):
            break
    return history
