"""BertAdam with the reference's constructor and update rule (method/optimization.py:223-343), executed as
ONE fused multi-tensor HIP step over a flat parameter buffer instead of ~10 elementwise launches per tensor.

At construction the parameters (and their .grad) are re-pointed into flat fp32 buffers (tensor starts aligned
to 256 elements); the data-parallel step all-reduces that flat gradient buffer as it is (dist.sync_gradients)."""
import math

import torch

from . import native

CHUNK = 256


def warmup_linear(progress, warmup):
    if progress < warmup:
        return progress / warmup
    return max((progress - 1.0) / (warmup - 1.0), 0.0)


def warmup_constant(progress, warmup):
    return progress / warmup if progress < warmup else 1.0


def warmup_cosine(progress, warmup, cycles=0.5):
    if progress < warmup:
        return progress / warmup
    progress = (progress - warmup) / (1 - warmup)
    return 0.5 * (1.0 + math.cos(math.pi * cycles * 2 * progress))


SCHEDULES = {None: None, "none": None, "warmup_linear": warmup_linear, "warmup_constant": warmup_constant,
             "warmup_cosine": warmup_cosine}


class FlatParams:
    """Owns flat (param, grad) buffers and re-points every parameter's storage into them.

    grad_buckets (data parallel): a list of parameter lists, in the order in which the backward pass completes them (train.
    backward_in_phases).  Every bucket then occupies ONE contiguous range of the flat buffers (bucket_ranges, element offsets),
    so it is all-reduced by one collective as soon as its last gradient exists; parameters in no bucket form a last range.
    Only the offsets change: every per-tensor array (t_start, t_numel, learning rates, weight decay) keeps the order of
    `params`."""

    def __init__(self, params, grad_buckets=None):
        self.params = [p for p in params]
        dev = self.params[0].device
        index = {id(p): i for i, p in enumerate(self.params)}
        bucket_of = [None] * len(self.params)
        n_b = 0
        for b, plist in enumerate(grad_buckets or []):
            for q in plist:
                i = index.get(id(q))
                if i is None:
                    raise ValueError("FlatParams: a grad bucket holds a parameter the optimizer does not own")
                if bucket_of[i] is not None:
                    raise ValueError("FlatParams: a parameter sits in two grad buckets")
                bucket_of[i] = b
            n_b = b + 1
        rest = [i for i, b in enumerate(bucket_of) if b is None]
        for i in rest:
            bucket_of[i] = n_b
        self.bucket_params = [[i for i, bb in enumerate(bucket_of) if bb == b] for b in range(n_b + (1 if rest else 0))]
        starts, numels, off = [0] * len(self.params), [p.numel() for p in self.params], 0
        layout, self.bucket_ranges = [], []
        for members in self.bucket_params:
            lo = off
            for i in members:
                starts[i] = off
                layout.append(i)
                off += (numels[i] + CHUNK - 1) // CHUNK * CHUNK
            self.bucket_ranges.append((lo, off))
        self.total = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        for p, s, n in zip(self.params, starts, numels):
            self.flat[s:s + n].copy_(p.data.reshape(-1))
            p.data = self.flat[s:s + n].view(p.shape)
            p.grad = self.grad[s:s + n].view(p.shape)
        chunk_tensor = []
        for t in layout:                                  # chunks follow the flat offsets, whatever the tensor order
            chunk_tensor += [t] * ((numels[t] + CHUNK - 1) // CHUNK)
        self.n_chunks = len(chunk_tensor)
        self.chunk_tensor = torch.tensor(chunk_tensor, dtype=torch.int32, device=dev)
        self.t_start = torch.tensor(starts, dtype=torch.int32, device=dev)
        self.t_numel = torch.tensor(numels, dtype=torch.int32, device=dev)
        self._starts, self._numels, self._views = starts, numels, None     # host copies: no .tolist() sync per step
        self._bound, self._had = False, tuple(True for _ in self.params)
        self._had_partial = {}
        self._had_subset = {}
        self._norm_done = set()
        self._index = index

    def views(self):
        if self._views is None:
            self._views = [self.grad[s:s + n].view(p.shape) for p, s, n in zip(self.params, self._starts, self._numels)]
        return self._views

    def drop_grads(self):
        """zero_grad(): leave every .grad None.  Autograd then ASSIGNS the gradient it computed instead of adding it into a
        pre-existing buffer - with .grad pointing into the zeroed flat buffer every backward pass ran one in-place add
        kernel per parameter (74 launches) plus a 23 MB fill."""
        for p in self.params:
            p.grad = None
        self._bound = False
        self._had_partial = {}
        self._had_subset = {}
        self._norm_done = set()

    FUSED_GATHER = True     # GPU: one native launch per gather (dldkd_gather_sumsq_f32) instead of torch's multi-tensor copy

    def _gather_native(self, members, norm2):
        """members' gradients -> their flat ranges (+ their sums of squares added to norm2) with dldkd_gather_sumsq_f32; False when a
        gradient is not a plain fp32 contiguous tensor of its parameter's size (the caller then copies the torch way)."""
        import ctypes
        views = self.views()
        todo = [i for i in members if self.params[i].grad is not None
                and (norm2 is not None or self.params[i].grad.data_ptr() != views[i].data_ptr())]
        for i in todo:
            g = self.params[i].grad
            if not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and g.numel() == self._numels[i]):
                return False
        L = native.lib()
        for lo in range(0, len(todo), 32):
            part = todo[lo:lo + 32]
            n = len(part)
            src = (ctypes.c_void_p * n)(*[self.params[i].grad.data_ptr() for i in part])
            st = (ctypes.c_int * n)(*[self._starts[i] for i in part])
            nu = (ctypes.c_int * n)(*[self._numels[i] for i in part])
            tt = (ctypes.c_int * n)(*part)
            native.check(L.dldkd_gather_sumsq_f32(src, st, nu, tt, n, native.ptr(self.grad), native.ptr(norm2), native.stream()),
                         "gather_sumsq")
        return True

    def _gather(self, members, norm2=None):
        """norm2 (per-tensor fp32 scratch, zeroed by the caller this step): the members' sums of squares are added to it in the same
        pass; self._norm_done then lists them (BertAdam.enqueue skips its own pass when every tensor is listed)."""
        views = self.views()
        if self.FUSED_GATHER and self.grad.is_cuda and self._gather_native(members, norm2):
            had = []
            for i in members:
                p = self.params[i]
                had.append(p.grad is not None)
                if p.grad is None:
                    views[i].zero_()
                p.grad = views[i]
            if norm2 is not None:
                self._norm_done.update(members)
            return had
        dst, src, had = [], [], []
        for i in members:
            p, view = self.params[i], views[i]
            had.append(p.grad is not None)
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr():
                dst.append(view)
                src.append(p.grad.detach().to(view.dtype).reshape(view.shape) if p.grad.dtype != view.dtype or p.grad.shape != view.shape
                           else p.grad.detach())
            p.grad = view
        if dst:
            torch._foreach_copy_(dst, src)
        return had

    def bind_views(self, had):
        """Point every p.grad at its flat view without copying (a replayed graph segment wrote the flat ranges itself)."""
        for p, v in zip(self.params, self.views()):
            p.grad = v
        self._bound, self._had, self._had_partial, self._had_subset = True, tuple(had), {}, {}
        self._norm_done = set()

    def gather_subset(self, params, norm2=None):
        """rebind_grads for an arbitrary subset of the parameters whose gradients are final (one tower's, on the stream that
        computed them: train.GraphedTrainStep captures a tower's backward pass and this copy into that tower's own graph).
        norm2: see _gather."""
        idx = [self._index[id(p)] for p in params]
        idx = [i for i in idx if i not in self._had_subset]
        for i, h in zip(idx, self._gather(idx, norm2)):
            self._had_subset[i] = h

    def norms_ready(self):
        """Every tensor's sum of squares was accumulated by this step's gathers."""
        return len(self._norm_done) == len(self.params)

    def rebind_bucket(self, b):
        """rebind_grads for the parameters of bucket b alone (their gradients are final: train.backward_in_phases)."""
        if b not in self._had_partial:
            self._had_partial[b] = self._gather(self.bucket_params[b])

    def rebind_grads(self):
        """Gather whatever autograd (or a caller) left in p.grad into the flat buffer with one multi-tensor copy, zero the
        views of parameters without a gradient, and re-point p.grad at the views.  Returns the tuple of per-parameter
        "had a gradient" flags (the reference's BertAdam skips parameters whose grad is None, optimization.py:294-295)."""
        views = self.views()
        if self._bound and all(p.grad is v for p, v in zip(self.params, views)):
            return self._had               # second call in one step (all-reduce, then optimizer): nothing new to gather
        had = [None] * len(self.params)
        todo = []
        for b, members in enumerate(self.bucket_params):
            if b in self._had_partial:     # gathered when the bucket completed
                for i, h in zip(members, self._had_partial[b]):
                    had[i] = h
            else:
                todo += members
        for i in todo:
            if i in self._had_subset:      # gathered with its tower (gather_subset)
                had[i] = self._had_subset[i]
        todo = [i for i in todo if i not in self._had_subset]
        for i, h in zip(todo, self._gather(todo)):
            had[i] = h
        self._bound, self._had = True, tuple(had)
        return self._had


class BertAdam(torch.optim.Optimizer):
    """Same arguments as the reference's BertAdam; `params` may be parameter groups with their own
    weight_decay (train.py:203-213)."""

    def __init__(self, params, lr, warmup=-1, t_total=-1, schedule="warmup_linear", b1=0.9, b2=0.999, e=1e-6,
                 weight_decay=0.01, max_grad_norm=1.0, grad_buckets=None, **kwargs):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if schedule not in SCHEDULES:
            raise ValueError("Invalid schedule parameter: {}".format(schedule))
        if not 0.0 <= b1 < 1.0 or not 0.0 <= b2 < 1.0 or not e >= 0.0:
            raise ValueError("Invalid b1 / b2 / e")
        defaults = dict(lr=lr, schedule=schedule, warmup=warmup, t_total=t_total, b1=b1, b2=b2, e=e,
                        weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        super().__init__(params, defaults)
        plist, wd, lrs = [], [], []
        for grp in self.param_groups:
            for p in grp["params"]:
                plist.append(p)
                wd.append(grp["weight_decay"])
                lrs.append(grp["lr"])
        self.fp = FlatParams(plist, grad_buckets)         # grad_buckets: see FlatParams (data-parallel overlap)
        dev = self.fp.flat.device
        self.m = torch.zeros_like(self.fp.flat)
        self.v = torch.zeros_like(self.fp.flat)
        self.t_wd = torch.tensor(wd, dtype=torch.float32, device=dev)
        self._base_lr = lrs
        self.t_lr = torch.zeros(len(plist), dtype=torch.float32, device=dev)
        # per-step host scalars reach the device through a ring of PINNED staging slots with asynchronous copies
        # (staging.PinnedRing): no pageable H2D, which blocks the host until the stream drains
        from .staging import PinnedRing
        self._lr_ring = PinnedRing(4 * len(plist), dev)
        self.t_active = torch.ones(len(plist), dtype=torch.float32, device=dev)
        self._active_key = tuple(True for _ in plist)
        self._base_lr_t = torch.tensor(lrs, dtype=torch.float32)
        self.norm2 = torch.zeros(len(plist), dtype=torch.float32, device=dev)
        self.step_count = 0

    def schedule_multiplier(self):
        g = self.param_groups[0]
        fn = SCHEDULES[g["schedule"]]
        if fn is None or g["t_total"] < 0:
            return 1.0
        return fn(float(self.step_count) / g["t_total"], g["warmup"])

    def get_lr(self):
        return [lr * self.schedule_multiplier() for lr in self._base_lr]

    def zero_grad(self, set_to_none=True):
        self.fp.drop_grads()

    def host_prepare(self, lr_out=None):
        """Host half of a step: this step's learning rates into `lr_out` (a CPU float32 tensor of one entry per parameter;
        default: the next slot of the optimizer's own pinned ring), step counter advanced.  train.GraphedTrainStep stages
        the rates together with the step's other host scalars and passes its own buffer."""
        if lr_out is None:
            lr_out = self._lr_ring.next()[:4 * len(self._base_lr)].view(torch.float32)
        torch.mul(self._base_lr_t, self.schedule_multiplier(), out=lr_out)
        self.step_count += 1

    @torch.no_grad()
    def zero_norms(self):
        """Zero the per-tensor norm scratch (the head of a step whose gathers accumulate the sums of squares: FlatParams._gather)."""
        native.check(native.lib().dldkd_zero_scratch_f32(native.ptr(self.norm2), self.norm2.numel(), native.stream()), "zero_scratch")

    @torch.no_grad()
    def enqueue(self, upload_lr=True):
        """Device half: gather the gradients into the flat buffer, upload the staged learning rates (asynchronous copy
        from pinned memory; upload_lr=False: the caller already put them into self.t_lr), one fused multi-tensor update.
        Enqueue-only, hence capturable into a hipGraph.  When this step's gathers already accumulated every tensor's sum of
        squares into self.norm2 (fp.norms_ready()) the update alone is launched."""
        norms_ready = self.fp.norms_ready()
        had = self.fp.rebind_grads()
        g = self.param_groups[0]
        if upload_lr:
            self._lr_ring.upload(self.t_lr.view(torch.uint8))
        if had != self._active_key:                      # rare: the set of parameters with a gradient changed
            self.t_active.copy_(torch.tensor(had, dtype=torch.float32))
            self._active_key = had
        L = native.lib()
        fn, what = (L.dldkd_bert_adam_update_f32, "bert_adam_update") if norms_ready else (L.dldkd_bert_adam_step_f32, "bert_adam_step")
        native.check(fn(native.ptr(self.fp.flat), native.ptr(self.fp.grad), native.ptr(self.m),
                                                native.ptr(self.v), native.ptr(self.fp.chunk_tensor), self.fp.n_chunks,
                                                native.ptr(self.fp.t_start), native.ptr(self.fp.t_numel), len(self.fp.params),
                                                native.ptr(self.norm2), native.ptr(self.t_wd), native.ptr(self.t_lr),
                                                native.ptr(self.t_active), g["b1"], g["b2"], g["e"], g["max_grad_norm"],
                                                native.stream()), what)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self.host_prepare()
        self.enqueue()
        from . import ops
        ops.bump_param_epoch()        # parameters changed through raw pointers: packed bf16 weight caches must repack
        return loss
