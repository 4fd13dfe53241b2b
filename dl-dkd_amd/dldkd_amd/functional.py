"""Differentiable ops: torch.autograd.Function shells whose forward AND backward are HIP kernels.

torch supplies tensors, the tape and the stream; every number is produced by libdldkd_hip.so.  Under
torch.no_grad() the same entry points run the fused inference kernels (ops.attention etc.).
"""
import contextlib
import math
import os

import torch
from torch.autograd import Function

from . import native, ops

HIDDEN, HEADS, DH = 384, 4, 96
_L = native.lib
_p, _s = native.ptr, native.stream


def _f32(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


# ---- zero arena: the ~45 small zero-initialised buffers of a training step (bias / LayerNorm / position-table gradient
# accumulators, loss gradients) carved from ONE buffer zeroed by one fill at the start of the step's forward pass, instead of one
# ATen fill kernel each (48 launches, 0.2 ms per C3 step; verdict r02).  A new buffer per step: nothing of the previous step is
# aliased; views keep it alive as long as any gradient still points into it.
_ARENA = None                 # [tensor, next offset (floats)]
ARENA_FLOATS = 1 << 19        # 2 MB


def begin_zero_arena(device):
    global _ARENA
    _ARENA = [torch.zeros(ARENA_FLOATS, dtype=torch.float32, device=device), 0]


def end_zero_arena():
    global _ARENA
    _ARENA = None


def _zeros(shape, device):
    """float32 zeros of `shape`: a slice of the step's arena when one is active on that device and has room, else torch.zeros."""
    a = _ARENA
    n = 1
    for d in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)):
        n *= int(d)
    if a is not None and a[0].device == torch.device(device) and n > 0:
        off = a[1]
        end = off + (n + 63) // 64 * 64               # 256-byte granules: every slice stays 16-byte aligned for the kernels
        if end <= ARENA_FLOATS:
            a[1] = end
            return a[0][off:off + n].view(shape)
    return torch.zeros(shape, dtype=torch.float32, device=device)


def _colsum(x2d, n_out):
    out = _zeros((n_out,), x2d.device)
    native.check(_L().dldkd_colsum_f32(_p(x2d), _p(out), x2d.shape[0], x2d.shape[1], _s()), "colsum")
    return out


def _axpy(a, b, alpha=1.0):
    native.check(_L().dldkd_axpy_f32(_p(a), _p(b), float(alpha), a.numel(), _s()), "axpy")
    return a


# ------------------------------------------------------------------------------------------ linear
class _Linear(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        ctx.rg = ops.row_groups(x.numel() // x.shape[-1])      # the tower's row-group flags (padding skipped) or None
        y = ops.linear(x, weight, bias, relu=relu, row_flags=ctx.rg)
        ctx.relu = relu
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @ops.in_backward
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        K, N = w.shape[1], w.shape[0]
        dy2 = _f32(dy).reshape(-1, N)
        if ctx.relu:
            dy2 = dy2.clone()
            native.check(_L().dldkd_relu_bwd_f32(_p(dy2), _p(y.reshape(-1, N)), dy2.numel(), _s()), "relu_bwd")
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        dx = ops.gemm(dy2, w, False, True, M, K, N, row_flags=ctx.rg).view(x.shape) if ctx.needs_input_grad[0] else None
        dw = ops.gemm(dy2, x2, True, True, N, K, M, row_flags=ctx.rg) if ctx.needs_input_grad[1] else None
        db = _colsum(dy2, N) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None


def linear(x, weight, bias=None, relu=False):
    x = _f32(x)
    if _needs_grad(x, weight, bias):
        return _Linear.apply(x, weight, bias, relu)
    return ops.linear(x, weight, bias, relu=relu)


# ------------------------------------------------------------------------------------------ layernorm
class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, add, add_mod):
        ctx.save_for_backward(x, gamma, add)
        ctx.add_mod = add_mod
        D = x.shape[-1]
        ctx.rg = ops.row_groups(x.numel() // D)
        if ctx.rg is not None:
            x2 = x.reshape(-1, D)
            out = torch.empty_like(x2)
            native.check(_L().dldkd_layernorm_groups_f32(_p(x2), _p(add), int(add_mod), _p(gamma), _p(beta), _p(out), None, x2.shape[0], D,
                                                         ops.LN_EPS, 0.0, 0, 0, None, _p(ctx.rg), _s()), "layernorm_groups")
            return out.view(x.shape)
        return ops.layernorm(x, gamma, beta, add=add, add_mod=add_mod)

    @staticmethod
    @ops.in_backward
    def backward(ctx, dy):
        x, gamma, add = ctx.saved_tensors
        D = x.shape[-1]
        dy = _f32(dy)
        need_dx = ctx.needs_input_grad[0] or (add is not None and ctx.needs_input_grad[3])
        dx = torch.empty_like(x) if need_dx else None
        dgb = _zeros((2, D), x.device)      # both accumulators
        dg, db = dgb[0], dgb[1]
        keep = ctx.keep if hasattr(ctx, "keep") else None
        rg = getattr(ctx, "rg", None)
        if rg is not None:
            native.check(_L().dldkd_layernorm_bwd_groups_f32(_p(x.reshape(-1, D)), _p(add), ctx.add_mod, _p(gamma), _p(dy.reshape(-1, D)),
                                                             _p(dx), _p(dg), _p(db), x.numel() // D, D, ops.LN_EPS, _p(keep),
                                                             getattr(ctx, "keep_scale", 1.0), _p(rg), _s()), "layernorm_bwd_groups")
        else:
            native.check(_L().dldkd_layernorm_bwd_f32(_p(x.reshape(-1, D)), _p(add), ctx.add_mod, _p(gamma), _p(dy.reshape(-1, D)),
                                                      _p(dx), _p(dg), _p(db), x.numel() // D, D, ops.LN_EPS, _p(keep),
                                                      getattr(ctx, "keep_scale", 1.0), _s()), "layernorm_bwd")
        dadd = None
        if add is not None and ctx.needs_input_grad[3]:
            if ctx.add_mod > 0:       # position table (L, D): sum over the batch
                rows = ctx.add_mod * D
                dadd = _colsum(dx.reshape(-1, rows), rows).view(ctx.add_mod, D)
            else:                     # residual: same gradient
                dadd = dx
        return (dx if ctx.needs_input_grad[0] else None), dg, db, dadd, None


class _LayerNormDropout(Function):
    """LayerNorm -> inverted dropout in one kernel forward (no normalised intermediate, no separate mask pass) and one kernel
    backward (dy masked on load inside the LayerNorm backward)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, add, add_mod, p):
        D = x.shape[-1]
        x2 = x.reshape(-1, D)
        out = torch.empty_like(x2)
        keep = torch.empty(x2.shape, dtype=torch.uint8, device=x.device)
        seed, off, state = _philox_slot(x.device, x.numel())
        ctx.rg = ops.row_groups(x2.shape[0])
        if ctx.rg is not None:
            native.check(_L().dldkd_layernorm_groups_f32(_p(x2), _p(add), int(add_mod), _p(gamma), _p(beta), _p(out), _p(keep), x2.shape[0], D,
                                                         ops.LN_EPS, float(p), seed, off, state, _p(ctx.rg), _s()), "layernorm_groups")
        else:
            native.check(_L().dldkd_layernorm_dropout_f32(_p(x2), _p(add), int(add_mod), _p(gamma), _p(beta), _p(out), _p(keep),
                                                          x2.shape[0], D, ops.LN_EPS, float(p), seed, off, state, _s()),
                         "layernorm_dropout")
        ctx.save_for_backward(x, gamma, add)
        ctx.add_mod, ctx.keep, ctx.keep_scale = add_mod, keep, 1.0 / (1.0 - p)
        return out.view(x.shape)

    @staticmethod
    @ops.in_backward
    def backward(ctx, dy):
        return _LayerNorm.backward(ctx, dy) + (None,)


def layernorm(x, gamma, beta, add=None, add_mod=0, p_drop=0.0, training=False):
    """LayerNorm(x [+ add]) [-> dropout(p_drop) when training]: the dropout is fused into the LayerNorm kernels."""
    x = _f32(x)
    add = _f32(add) if add is not None else None
    if training and p_drop > 0.0:
        return _LayerNormDropout.apply(x, gamma, beta, add, add_mod, float(p_drop))
    if _needs_grad(x, gamma, beta, add):
        return _LayerNorm.apply(x, gamma, beta, add, add_mod)
    return ops.layernorm(x, gamma, beta, add=add, add_mod=add_mod)


# ---- both video towers' input-projection LayerNorm in one pass (round 6).  The inheritance and the exploration tower normalise the
# SAME student features (method/model.py:229-243): DLDKD.forward_tensors calls in_proj_ln_dual in front of the towers; the two
# _InProjTrain.forward calls that follow find their rows here (matched by the raw rows' storage and the branch's gamma) and skip
# their own LayerNorm launch - the 201-MB TVR batch is read once instead of twice.
# ---- two-plane operands of the "mixed" precision's forward GEMMs (round 6): an fp32 tensor as [2][rows][K] bf16 (h = bf16(x), m =
# bf16(x - h)); dldkd_gemm_bf16_nt16_planes contracts (m, h) + (h, m) + (h, h) in one pass of the LDS-DMA bf16 kernel - the numbers of
# gemm_f32x2 (in-kernel split, 128 x 128 x 16 tiles) at 2.5x its rate.  DLDKD_MIXED_PLANES=0: the in-kernel-split kernel (A/B).
MIXED_PLANES = os.environ.get("DLDKD_MIXED_PLANES", "1") == "1"


def mixed_planes_ok(M, N, K):
    return bool(MIXED_PLANES and ops.gemm_precision() == "fp32x2" and _L().dldkd_gemm_bf16_nt16_ok(int(M), int(N), int(K), int(K), int(K)))


def split2_jobs(jobs, device):
    """jobs = [(src fp32 contiguous, "split" | "copy")] -> their outputs from ONE launch: split -> bf16 (2,) + src.shape, copy -> a
    fp32 tensor of src's shape; sources of one `group` tuple ((srcs...), "split") are concatenated along dim 0 first."""
    import ctypes
    outs, flat = [], []
    for src, kind in jobs:
        group = src if isinstance(src, (tuple, list)) else (src,)
        rows = sum(t.shape[0] for t in group)
        tail = tuple(group[0].shape[1:])
        n_tot = rows
        for d_ in tail:
            n_tot *= int(d_)
        if kind == "split":
            out = torch.empty((2, rows) + tail, dtype=torch.bfloat16, device=device)
        else:
            out = torch.empty((rows,) + tail, dtype=torch.float32, device=device)
        off = 0
        for t in group:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() % 4:
                raise native.NativeError("split2_jobs: sources must be contiguous fp32 with a multiple of 4 elements")
            n = t.numel()
            if kind == "split":
                flat.append((t.data_ptr(), out.data_ptr() + 2 * off, out.data_ptr() + 2 * (n_tot + off), n, 0))
            else:
                flat.append((t.data_ptr(), out.data_ptr() + 4 * off, 0, n, 1))
            off += n
        outs.append(out)
    nj = len(flat)
    if nj > 12:
        raise native.NativeError("split2_jobs: at most 12 jobs per launch")
    src = (ctypes.c_void_p * nj)(*[f[0] for f in flat])
    dh = (ctypes.c_void_p * nj)(*[f[1] for f in flat])
    dm = (ctypes.c_void_p * nj)(*[f[2] for f in flat])
    nn = (ctypes.c_long * nj)(*[f[3] for f in flat])
    kk = (ctypes.c_int * nj)(*[f[4] for f in flat])
    native.check(_L().dldkd_split2_bf16_jobs(src, dh, dm, nn, kk, nj, _s()), "split2_bf16_jobs")
    return outs


def gemm_planes(ap, bp, bias, M, N, K, relu=False, row_flags=None):
    """y (M, N) fp32 = act(A B^T + bias) from the two-plane operands ap (2, M, K) and bp (2, N, K)."""
    y = torch.empty(M, N, dtype=torch.float32, device=ap.device)
    native.check(_L().dldkd_gemm_bf16_nt16_planes(_p(ap), _p(bp), _p(bias), _p(y), M, N, K, K, K, N, int(relu), _p(row_flags), M * K, N * K, _s()),
                 "gemm_bf16_nt16_planes")
    return y


IN_PROJ_LN_DUAL = os.environ.get("DLDKD_LN_DUAL", "1") == "1"
_PRE_LN = {}


def in_proj_ln_dual_ok(x, layers, p, training):
    """Throughput (bf16) training mode, two LinearLayers over one contiguous fp32 GPU tensor of raw features that needs no gradient,
    the one-GEMM backward (no keep bytes), the bf16 x bf16 forward GEMM."""
    if not (IN_PROJ_LN_DUAL and IN_PROJ_TRAIN_BF16_ROWS and IN_PROJ_TRAIN_FUSED and IN_PROJ_BWD_DUAL and IN_PROJ_TRAIN_NT16
            and not IN_PROJ_KEEP_BYTES and IN_PROJ_SKIP_PADDING and len(layers) == 2 and training and torch.is_grad_enabled()):
        return False
    mixed = ops.precision_mode() == "mixed" and ops.gemm_precision() == "fp32x2" and MIXED_PLANES     # (rows as two bf16 planes)
    if not ((ops.gemm_precision() == "bf16" or mixed) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3
            and not x.requires_grad and x.shape[1] % 32 == 0):
        return False
    K = x.shape[-1]
    M = x.shape[0] * x.shape[1]
    for l in layers:
        w = l.net[1].weight
        N = w.shape[0]
        if not (l.relu and l.layer_norm and w.requires_grad and l.LayerNorm.weight.requires_grad and l.LayerNorm.bias.requires_grad
                and w.is_contiguous() and N <= 384 and N % 2 == 0 and K % 4 == 0 and K <= 4096 and M >= 1024
                and _L().dldkd_gemm_bf16_nt16_ok(M, N, K, K, K)):
            return False
    return True


_PRE_SLOT = {}


def predraw_in_proj_slots(x, layers, p):
    """The dropout slots of the two video towers' input projections, drawn up front, branch 0 first (DLDKD.forward_tensors, in
    front of the towers, in EVERY precision mode and with or without the one-pass LayerNorm): the masks of a step do not depend on
    which of those forms runs - the tests that compare two forms of the step with dropout on rely on it."""
    _PRE_SLOT.clear()
    if p > 0.0:
        x2 = x.reshape(-1, x.shape[-1])
        for l in layers:
            _PRE_SLOT[(x2.data_ptr(), l.LayerNorm.weight.data_ptr())] = _philox_slot(x.device, x.numel())


def _in_proj_slot(x2, gamma, n):
    s = _PRE_SLOT.pop((x2.data_ptr(), gamma.data_ptr()), None)
    return s if s is not None else _philox_slot(x2.device, n)


def in_proj_ln_dual(x, row_mask, layers, p):
    """One launch: z_b = dropout(LayerNorm(x; gamma_b, beta_b)) as bf16 rows for both layers, shared statistics and group flags.
    The Philox slots are drawn here, branch 0 first."""
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    M, dev = x2.shape[0], x.device
    planes = ops.gemm_precision() == "fp32x2"              # "mixed": [2][M][K] per branch, plane 0 = the bf16 rows
    z = [torch.empty((2, M, K) if planes else (M, K), dtype=torch.bfloat16, device=dev) for _ in layers]
    stats = torch.empty(2, M, dtype=torch.float32, device=dev)
    gflags = torch.empty(M // 32, dtype=torch.uint8, device=dev)
    rm = _f32(row_mask).reshape(-1)
    g = [l.LayerNorm.weight for l in layers]
    slots = [_in_proj_slot(x2, gi, x.numel()) if p > 0.0 else (0, 0, None) for gi in g]
    b = [l.LayerNorm.bias for l in layers]
    native.check(_L().dldkd_layernorm_dropout_bf16_dual(_p(x2), _p(g[0]), _p(b[0]), _p(g[1]), _p(b[1]), _p(z[0]), _p(z[1]), _p(stats), M, K,
                                                        ops.LN_EPS, float(p), slots[0][0], slots[0][1], slots[1][1], slots[0][2], _p(rm),
                                                        _p(gflags), int(planes), _s()), "layernorm_dropout_bf16_dual")
    _PRE_LN.clear()
    here = torch.cuda.current_stream(dev)
    for i in range(2):
        _PRE_LN[(x2.data_ptr(), g[i].data_ptr())] = (z[i], stats, gflags, slots[i], float(p), M, K, here)


def _take_pre_ln(x2, gamma, p, row_mask):
    ent = _PRE_LN.pop((x2.data_ptr(), gamma.data_ptr()), None)
    if ent is None:
        return None
    z, stats, gflags, slot, p0, M, K, made_on = ent
    if p0 != float(p) or (M, K) != tuple(x2.shape) or row_mask is None:
        return None
    cur = torch.cuda.current_stream(z.device)
    if cur != made_on:
        # the rows were allocated on the stream of in_proj_ln_dual and are read - forward GEMM now, weight gradient in the backward pass -
        # on THIS tower's stream: without this the allocator hands their memory to the producer's stream the moment the backward pass
        # drops them, while this stream may not have run its reads yet (inside ONE captured graph with forked tower streams that
        # order is real: the single-graph stepper's dW came out as NaN at the TVR batch)
        for t in (z, stats, gflags):
            t.record_stream(cur)
    return z, stats, gflags, slot


def drop_pre_ln():
    _PRE_LN.clear()
    _PRE_SLOT.clear()


class _InProjTrain(Function):
    """LinearLayer on RAW features in training, throughput mode (model_components.py:294-312): LayerNorm -> Dropout -> Linear ->
    ReLU as one autograd node.  The features need no gradient, so the backward pass never forms the Linear's input gradient: the
    LayerNorm parameter gradients come out of the accumulators of dy W (dldkd_linear_lngrad) - one GEMM-with-epilogue
    instead of the dX GEMM (201 MB written at the TVR batch) plus a LayerNorm backward pass over x and that gradient."""

    @staticmethod
    def forward(ctx, x, gamma, beta, weight, bias, p, relu, row_mask=None, grad_premasked=False):
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        M, N = x2.shape[0], weight.shape[0]
        keep = stats = gflags = None
        # throughput mode: the LayerNorm-dropout rows are WRITTEN as bf16 - what the bf16 GEMMs round them to anyway - so the
        # forward GEMM and dW read half the bytes (201 -> 100 MB per branch at the TVR batch), and the row statistics are kept
        # for the backward pass instead of being recomputed from x there
        z16 = IN_PROJ_TRAIN_BF16_ROWS and ops.gemm_precision() == "bf16" and K % 4 == 0 and N % 2 == 0
        pre = _take_pre_ln(x2, gamma, p, row_mask) if z16 else None
        mixed16 = (ops.precision_mode() == "mixed" and IN_PROJ_BWD_DUAL and not IN_PROJ_KEEP_BYTES and K % 4 == 0 and N <= 384 and N % 2 == 0
                   and weight.is_contiguous() and gamma.requires_grad and beta.requires_grad and weight.requires_grad)
        if pre is not None:
            # both branches' LayerNorm-dropout rows were written by ONE pass over the raw features (in_proj_ln_dual, called by the
            # model in front of the towers): this branch's rows, the shared statistics and group flags, its Philox slot
            z, stats, gflags, (seed, off, state) = pre
            y = torch.empty(M, N, dtype=torch.float32, device=x.device)
            self_w16 = _take_prepacked("w16", weight.data_ptr())
            if self_w16 is None:
                self_w16 = torch.empty(N, K, dtype=torch.bfloat16, device=x.device)
                native.check(_L().dldkd_cast_bf16(_p(weight), _p(self_w16), N * K, _s()), "cast_bf16")
            native.check(_L().dldkd_gemm_bf16_nt16(_p(z), _p(self_w16), _p(bias), _p(y), M, N, K, K, K, N, int(relu), _p(gflags), _s()),
                         "gemm_bf16_nt16")
        elif mixed16:
            # "mixed" precision: the forward product on the fp32-grade GEMM over fp32 LayerNorm-dropout rows (exact loss values); the
            # SAME launch leaves the rows as bf16 too - all the backward pass reads (the one-GEMM bf16 backward below: dW and the
            # LayerNorm parameter gradients; the fp32 rows die with this call)
            stats = torch.empty(2, M, dtype=torch.float32, device=x.device)
            rm = None
            if row_mask is not None and IN_PROJ_SKIP_PADDING and x.dim() == 3 and row_mask.numel() == M:
                rm = _f32(row_mask).reshape(-1)
                if x.shape[1] % 32 == 0:          # (query towers: 30 words - no 32-row groups; the rows of the padding are still skipped)
                    gflags = torch.empty(M // 32, dtype=torch.uint8, device=x.device)
            pre2 = _take_pre_ln(x2, gamma, p, row_mask) if mixed_planes_ok(M, N, K) else None
            if pre2 is None or pre2[0].dim() != 3:
                pre2 = None
                seed, off, state = _in_proj_slot(x2, gamma, x.numel()) if p > 0.0 else (0, 0, None)
            if pre2 is not None:
                # both branches' planes came from ONE pass over the raw features (in_proj_ln_dual in front of the towers)
                zp, stats, gflags, (seed, off, state) = pre2
                wp = _take_prepacked("w_in_planes", weight.data_ptr())
                if wp is None:
                    (wp,) = split2_jobs([(weight, "split")], x.device)
                y = gemm_planes(zp, wp, bias, M, N, K, relu=relu, row_flags=gflags)
                z = zp[0]
            elif mixed_planes_ok(M, N, K):
                # the LayerNorm-dropout rows as TWO bf16 planes (same bytes as fp32 rows): the forward GEMM contracts them as they lie
                # (LDS-DMA, no split on the way), and plane 0 IS the bf16 row the backward pass reads
                zp = torch.empty(2, M, K, dtype=torch.bfloat16, device=x.device)
                native.check(_L().dldkd_layernorm_ex_f32(_p(x2), None, 0, _p(gamma), _p(beta), None, None, _p(zp), None, _p(stats), M, K, ops.LN_EPS,
                                                         float(p), seed, off, state, _p(rm), _p(gflags), None, _s()), "layernorm_ex")
                wp = _take_prepacked("w_in_planes", weight.data_ptr())
                if wp is None:
                    (wp,) = split2_jobs([(weight, "split")], x.device)
                y = gemm_planes(zp, wp, bias, M, N, K, relu=relu, row_flags=gflags)
                z = zp[0]
            else:
                zf = torch.empty_like(x2)
                z = torch.empty(x2.shape, dtype=torch.bfloat16, device=x.device)
                native.check(_L().dldkd_layernorm_ex_f32(_p(x2), None, 0, _p(gamma), _p(beta), _p(zf), _p(z), None, None, _p(stats), M, K, ops.LN_EPS,
                                                         float(p), seed, off, state, _p(rm), _p(gflags), None, _s()), "layernorm_ex")
                y = ops.linear(zf, weight, bias, relu=relu, row_flags=gflags)
                del zf
        elif z16:
            z = torch.empty(x2.shape, dtype=torch.bfloat16, device=x.device)
            stats = torch.empty(2, M, dtype=torch.float32, device=x.device)
            seed, off, state = (0, 0, None)
            if p > 0.0:
                # the keep bytes (one per element: 50 MB per video tower at the TVR batch) are written only where the backward pass
                # will read them - the two-GEMM fallback; the one-GEMM backward (IN_PROJ_BWD_DUAL) draws the few bits its
                # small-gamma columns need again from the Philox slot saved below
                dual = (IN_PROJ_BWD_DUAL and not IN_PROJ_KEEP_BYTES and gamma.requires_grad and beta.requires_grad and weight.requires_grad
                        and N <= 384 and weight.is_contiguous())
                if not dual:
                    keep = torch.empty(x2.shape, dtype=torch.uint8, device=x.device)
                seed, off, state = _in_proj_slot(x2, gamma, x.numel())
            # rows of the padding (row_mask == 0) are never read and come out as zero rows; when the padded length is a multiple of
            # 32 the kernel also flags the 32-row groups that hold valid rows, and dW below skips the others
            rm = None
            if row_mask is not None and IN_PROJ_SKIP_PADDING and x.dim() == 3 and row_mask.numel() == M:
                rm = _f32(row_mask).reshape(-1)
                if x.shape[1] % 32 == 0:
                    gflags = torch.empty(M // 32, dtype=torch.uint8, device=x.device)
            native.check(_L().dldkd_layernorm_dropout_bf16(_p(x2), _p(gamma), _p(beta), _p(z), _p(keep), _p(stats), M, K, ops.LN_EPS,
                                                           float(p), seed, off, state, _p(rm), _p(gflags), _s()), "layernorm_dropout_bf16")
            y = torch.empty(M, N, dtype=torch.float32, device=x.device)
            if IN_PROJ_TRAIN_NT16 and M >= 1024 and _L().dldkd_gemm_bf16_nt16_ok(M, N, K, K, K) and weight.is_contiguous():
                # both operands bf16, tiles by LDS-DMA (gemm_bf16_dma.hip): the weight is cast once here (1.2 M elements)
                w16 = _take_prepacked("w16", weight.data_ptr())
                if w16 is None:
                    w16 = torch.empty(N, K, dtype=torch.bfloat16, device=x.device)
                    native.check(_L().dldkd_cast_bf16(_p(weight), _p(w16), N * K, _s()), "cast_bf16")
                native.check(_L().dldkd_gemm_bf16_nt16(_p(z), _p(w16), _p(bias), _p(y), M, N, K, K, K, N, int(relu), _p(gflags), _s()),
                             "gemm_bf16_nt16")
            else:
                native.check(_L().dldkd_gemm_bf16_mixed(0, _p(z), _p(weight), _p(bias), _p(y), M, N, K, K, K, N, int(relu), None, 0,
                                                        _p(gflags), _s()), "gemm_bf16_mixed")
        elif (row_mask is not None and IN_PROJ_SKIP_PADDING and x.dim() == 3 and row_mask.numel() == M and x.shape[1] % 32 == 0
              and M % 128 == 0 and ops.gemm_precision() in ("fp32", "fp32x3", "fp32x2")):
            # parity mode with the batch's mask: the same padding skip with fp32 rows (the three-plane GEMMs are compute-bound:
            # the 32-row groups that are not multiplied are time saved one for one)
            z = torch.empty_like(x2)
            stats = torch.empty(2, M, dtype=torch.float32, device=x.device)
            gflags = torch.empty(M // 32, dtype=torch.uint8, device=x.device)
            seed, off, state = (0, 0, None)
            if p > 0.0:
                keep = torch.empty(x2.shape, dtype=torch.uint8, device=x.device)
                seed, off, state = _in_proj_slot(x2, gamma, x.numel())
            native.check(_L().dldkd_layernorm_dropout_rows_f32(_p(x2), _p(gamma), _p(beta), _p(z), _p(keep), _p(stats), M, K, ops.LN_EPS,
                                                               float(p), seed, off, state, _p(_f32(row_mask).reshape(-1)), _p(gflags),
                                                               _s()), "layernorm_dropout_rows")
            y = ops.linear(z, weight, bias, relu=relu, row_flags=gflags)
        else:
            if p > 0.0:
                z = torch.empty_like(x2)
                keep = torch.empty(x2.shape, dtype=torch.uint8, device=x.device)
                seed, off, state = _in_proj_slot(x2, gamma, x.numel())
                native.check(_L().dldkd_layernorm_dropout_f32(_p(x2), None, 0, _p(gamma), _p(beta), _p(z), _p(keep), M, K,
                                                              ops.LN_EPS, float(p), seed, off, state, _s()), "layernorm_dropout")
            else:
                z = ops.layernorm(x2, gamma, beta)
            y = ops.linear(z, weight, bias, relu=relu)
        global _LAST_GROUP_FLAGS
        _LAST_GROUP_FLAGS = (gflags, M) if gflags is not None else None
        # grad_premasked: the node behind this one (the fused training tower) hands back a gradient that already carries the ReLU mask
        # [y > 0] - it reads y anyway - so the backward pass here neither keeps y nor runs relu_bwd over a clone of dy
        ctx.save_for_backward(x2, weight, z, y if (relu and not grad_premasked) else None, keep, stats, gflags, gamma, beta)
        ctx.relu, ctx.has_bias, ctx.keep_scale, ctx.prec = (relu and not grad_premasked), bias is not None, 1.0 / (1.0 - p), ops.gemm_precision()
        # the dropout's Philox slot, for a backward pass that redraws bits instead of reading keep bytes (z16 path without `keep`)
        ctx.drop_rng = ((float(p), seed, off, state, _philox_step.dev if _philox_step is not None else None)
                        if ((z16 or mixed16) and p > 0.0) else (0.0, 0, 0, None, None))
        return y.view(*x.shape[:-1], N)

    @staticmethod
    @ops.in_backward
    def backward(ctx, dy):
        x2, w, z, y, keep, stats, gflags, gamma, beta = ctx.saved_tensors
        N, K = w.shape
        M = x2.shape[0]
        dy2 = _f32(dy).reshape(-1, N)
        if ctx.relu:
            dy2 = dy2.clone()
            native.check(_L().dldkd_relu_bwd_f32(_p(dy2), _p(y.reshape(-1, N)), dy2.numel(), _s()), "relu_bwd")
        dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[4]
        if (IN_PROJ_BWD_DUAL and z.dtype == torch.bfloat16 and ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and ctx.needs_input_grad[3]
                and N <= 384 and N % 2 == 0 and w.is_contiguous()):
            # ONE weight-gradient GEMM with two accumulator sets gives dW and the LayerNorm parameter gradients (the (M, K) product
            # dy W of dldkd_linear_lngrad reassociated into dW's M-long contraction: gemm_bf16.hip, gemm_bf16_dw_dual_kernel)
            dw = torch.empty(N, K, dtype=torch.float32, device=x2.device)
            db = _zeros((N,), x2.device) if want_db else None
            dgb = _zeros((2, K), x2.device)
            nbytes = _L().dldkd_inproj_bwd_workspace_bytes(N, K, M)
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x2.device)
            p_, seed_, off_, state_, _alive = ctx.drop_rng
            # the bf16 copy of dy the fused tower's last backward kernel left beside the fp32 rows (else the library casts them)
            dy16 = _DY16.pop(dy.data_ptr(), None) if (not ctx.relu and dy.dtype == torch.float32 and dy.is_contiguous()) else None
            if dy16 is not None and dy16.numel() != dy2.numel():
                dy16 = None
            native.check(_L().dldkd_inproj_bwd_bf16(_p(dy2), _p(z), _p(w), _p(gamma), _p(beta), ctx.keep_scale, _p(x2), _p(keep),
                                                    p_, seed_, off_, state_, _p(stats[0]), _p(stats[1]), _p(dw), _p(db), _p(dgb[0]), _p(dgb[1]),
                                                    M, N, K, _p(ws), nbytes, _p(gflags), _p(dy16), _s()), "inproj_bwd_bf16")
            return None, dgb[0], dgb[1], dw, db, None, None, None, None
        if keep is None and ctx.drop_rng[0] > 0.0:
            # the forward pass wrote no keep bytes because every parameter of the layer wanted a gradient (the one-GEMM backward
            # redraws the bits); a backward pass that asks for a subset lands here, where the two-GEMM path reads the bytes
            raise RuntimeError("in_proj_train: this backward pass needs the dropout keep bytes the forward pass did not write "
                               "(gradients requested for a subset of gamma / beta / weight): set functional.IN_PROJ_KEEP_BYTES = True")
        if ctx.needs_input_grad[3]:
            if z.dtype == torch.bfloat16:
                dw = torch.empty(N, K, dtype=torch.float32, device=x2.device)
                ws, ws_bytes = ops._gemm_workspace(_L(), N, K, M, True, True, x2.device, precision="bf16")
                if want_db:         # the bias gradient = the column sums of dy: taken from the dy tiles the weight-gradient GEMM stages anyway
                    db = _zeros((N,), x2.device)
                native.check(_L().dldkd_gemm_bf16_dw_bias(1, _p(dy2), _p(z), _p(dw), N, K, M, N, K, _p(ws), ws_bytes, _p(gflags), _p(db),
                                                          _s()), "gemm_bf16_dw_bias")
            else:
                dw = ops.gemm(dy2, z, True, True, N, K, M, row_flags=gflags)
        if want_db and db is None:
            db = _colsum(dy2, N)
        dg = dbeta = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            if stats is None:
                stats = torch.empty(2, M, dtype=torch.float32, device=x2.device)
                native.check(_L().dldkd_row_meanrstd_f32(_p(x2), _p(stats[0]), _p(stats[1]), M, K, ops.LN_EPS, _s()), "row_meanrstd")
            tiles = (M + 127) // 128
            ws = torch.empty(2 * tiles * K, dtype=torch.float32, device=x2.device)
            dgb = _zeros((2, K), x2.device)
            native.check(_L().dldkd_linear_lngrad(ops._PREC_ID[ops.gemm_precision()] if ops.gemm_precision() != "fp32x2" else 1, _p(dy2), _p(w), _p(x2), _p(keep), ctx.keep_scale, _p(stats[0]), _p(stats[1]),
                                                       _p(ws), ws.numel() * 4, _p(dgb[0]), _p(dgb[1]), M, N, K, _p(gflags), _s()), "linear_lngrad")
            dg, dbeta = dgb[0], dgb[1]
        return None, dg, dbeta, dw, db, None, None, None, None


_LAST_GROUP_FLAGS = None


def take_group_flags():
    """(flags, M) of the input projection that just ran with a row mask (None otherwise); cleared by the call."""
    global _LAST_GROUP_FLAGS
    r, _LAST_GROUP_FLAGS = _LAST_GROUP_FLAGS, None
    return r


IN_PROJ_SKIP_PADDING = True           # ... and the rows of the padding (a row mask given) are neither normalised nor multiplied
IN_PROJ_TRAIN_BF16_ROWS = True        # throughput mode: the saved LayerNorm-dropout rows of the input projection are bf16
IN_PROJ_TRAIN_FUSED = True
IN_PROJ_BWD_DUAL = True               # throughput mode: dW and the LayerNorm parameter gradients from one two-accumulator GEMM
IN_PROJ_KEEP_BYTES = False            # ... True: the forward pass still writes a keep byte per element (A/B and fallback)
IN_PROJ_TRAIN_NT16 = True             # throughput mode: the forward GEMM of the input projection on the bf16 x bf16 LDS-DMA kernel


def in_proj_train_ok(x, weight):
    """Training, features without a gradient, a tiled-GEMM precision mode (bf16 or the three-plane fp32-grade one), row statistics
    kernel limits (D % 4 == 0, D <= 4096)."""
    return (IN_PROJ_TRAIN_FUSED and ops.gemm_precision() in ("bf16", "fp32", "fp32x3", "fp32x2") and x.is_cuda and torch.is_grad_enabled() and not x.requires_grad
            and x.shape[-1] % 4 == 0 and x.shape[-1] <= 4096 and weight.requires_grad)


def in_proj_train(x, gamma, beta, weight, bias, p_drop, training, relu=True, row_mask=None, grad_premasked=False):
    return _InProjTrain.apply(_f32(x), gamma, beta, weight, bias, float(p_drop) if training else 0.0, bool(relu), row_mask,
                              bool(grad_premasked))


# ------------------------------------------------------------------------------------------ dropout
class PhiloxStepState:
    """Philox (seed, base offset) of ONE training step in device memory, for hipGraph-captured steps (train.GraphedTrainStep).

    Eager dropout calls read (seed, offset) from torch's CUDA generator on the host and pass them as kernel arguments; a
    captured graph would replay the same numbers.  While a PhiloxStepState is active (capture), every dropout call instead
    bakes only its RELATIVE offset inside the step into the graph and the kernel adds the base it finds in `dev` (two int64
    words of the step's staged-scalars buffer).  Before each replay begin_step() takes (seed, offset) from torch's
    generator, advances the generator by what the step consumes and writes them into the host staging slot.  The masks of
    a captured run are therefore exactly those of the eager run with the same torch.manual_seed."""

    def __init__(self, dev_words):
        self.dev = dev_words             # int64[2] device view
        self.consumed = 0                # offsets one step draws (fixed once captured)

    def slot(self, n):
        rel = self.consumed
        self.consumed += 4 * ((n + 3) // 4)
        return rel

    def begin_step(self, host_words):
        dev = self.dev.device
        gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
        off = gen.get_offset()
        gen.set_offset(off + self.consumed)
        seed = gen.initial_seed() & 0xFFFFFFFFFFFFFFFF
        host_words[0] = seed - (1 << 64) if seed >= (1 << 63) else seed          # two's complement into int64
        host_words[1] = off


_philox_step = None        # the active PhiloxStepState (only while a training step is being captured)


def set_philox_step(state):
    global _philox_step
    old, _philox_step = _philox_step, state
    return old


def _philox_slot(device, n):
    """(seed, offset, state pointer) for n elements.  Eager: from torch's CUDA generator, advanced like a torch op that
    draws n numbers would, so torch.manual_seed / get_rng_state / set_rng_state govern our masks too.  Under capture: the
    step-relative offset and the device state the kernel adds to it (PhiloxStepState)."""
    if _philox_step is not None:
        return 0, _philox_step.slot(n), _p(_philox_step.dev)
    gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
    off = gen.get_offset()
    gen.set_offset(off + 4 * ((n + 3) // 4))          # torch offsets move in multiples of 4
    return gen.initial_seed() & 0xFFFFFFFFFFFFFFFF, off, None


def _dropout_fwd(x, p):
    out = torch.empty_like(x)
    keep = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    seed, off, state = _philox_slot(x.device, x.numel())
    native.check(_L().dldkd_dropout_fwd_f32(_p(x), _p(out), _p(keep), x.numel(), float(p), seed, off, state, _s()), "dropout_fwd")
    return out, keep


class _Dropout(Function):
    @staticmethod
    def forward(ctx, x, p):
        out, keep = _dropout_fwd(x, p)
        ctx.save_for_backward(keep)
        ctx.scale = 1.0 / (1.0 - p)
        return out

    @staticmethod
    @ops.in_backward
    def backward(ctx, dy):
        (keep,) = ctx.saved_tensors
        dy = _f32(dy)
        out = torch.empty_like(dy)
        native.check(_L().dldkd_mask_scale_f32(_p(dy), _p(keep), ctx.scale, _p(out), dy.numel(), _s()), "mask_scale")
        return out, None


def dropout(x, p, training):
    """Inverted dropout: Philox mask + scale in one kernel (seed/offset from torch's CUDA generator)."""
    if not training or p <= 0.0:
        return x
    return _Dropout.apply(_f32(x), float(p))


# ------------------------------------------------------------------------------------------ attention
class _AttentionTrain(Function):
    """Training form of BertSelfAttention (model_components.py:398-436), fused (attention_train.hip): one forward kernel
    (probabilities saved, dropout in registers) and two backward kernels instead of six batched GEMMs + row softmax
    fwd/bwd + dropout fwd/bwd.  qkv (N, L, 1152); returns the context layer (N, L, 384)."""

    @staticmethod
    def forward(ctx, qkv, mask, p_drop):
        N, L = qkv.shape[0], qkv.shape[1]
        P = torch.empty(N, HEADS, L, L, dtype=torch.float32, device=qkv.device)
        out = torch.empty(N, L, HIDDEN, dtype=torch.float32, device=qkv.device)
        seed, off, state = _philox_slot(qkv.device, P.numel()) if p_drop > 0.0 else (0, 0, None)
        native.check(_L().dldkd_attention_train_fwd_f32(_p(qkv), _p(mask), _p(P), _p(out), N, L, float(p_drop), seed, off, state,
                                                        _s()), "attention_train_fwd")
        ctx.save_for_backward(qkv, P)
        ctx.rng = (float(p_drop), seed, off, state, _philox_step.dev if _philox_step is not None else None)
        return out

    @staticmethod
    @ops.in_backward
    def backward(ctx, dout):
        qkv, P = ctx.saved_tensors
        p_drop, seed, off, state, _keep_alive = ctx.rng
        N, L = qkv.shape[0], qkv.shape[1]
        dqkv = torch.empty_like(qkv)
        dS = torch.empty_like(P)
        native.check(_L().dldkd_attention_train_bwd_f32(_p(qkv), _p(_f32(dout)), _p(P), _p(dS), _p(dqkv), N, L, p_drop, seed, off,
                                                        state, _s()), "attention_train_bwd")
        return dqkv, None, None


class _AttentionTrainBf16(Function):
    """Throughput-mode twin (attention_train_bf16.hip): every product on the bf16 matrix cores; nothing saved but qkv and the
    mask - the one backward kernel recomputes the probabilities (same Philox slot, hence the same keep bits)."""

    @staticmethod
    def forward(ctx, qkv, mask, p_drop):
        N, L = qkv.shape[0], qkv.shape[1]
        out = torch.empty(N, L, HIDDEN, dtype=torch.float32, device=qkv.device)
        seed, off, state = _philox_slot(qkv.device, N * HEADS * L * L) if p_drop > 0.0 else (0, 0, None)
        native.check(_L().dldkd_attention_train_fwd_bf16(_p(qkv), _p(mask), _p(out), N, L, float(p_drop), seed, off, state, _s()),
                     "attention_train_fwd_bf16")
        ctx.save_for_backward(qkv, mask)
        ctx.rng = (float(p_drop), seed, off, state, _philox_step.dev if _philox_step is not None else None)
        return out

    @staticmethod
    @ops.in_backward
    def backward(ctx, dout):
        qkv, mask = ctx.saved_tensors
        p_drop, seed, off, state, _keep_alive = ctx.rng
        N, L = qkv.shape[0], qkv.shape[1]
        dqkv = torch.empty_like(qkv)
        native.check(_L().dldkd_attention_train_bwd_bf16(_p(qkv), _p(mask), _p(_f32(dout)), _p(dqkv), N, L, p_drop, seed, off, state,
                                                         _s()), "attention_train_bwd_bf16")
        return dqkv, None, None


ATTN_TRAIN_BF16 = True      # throughput mode: bf16-MFMA training attention (False: the exact-fp32 kernels in both modes)


def attention(qkv, mask, p_drop=0.0, training=False):
    qkv = _f32(qkv)
    mask = _f32(mask) if mask is not None else None
    if _needs_grad(qkv) or (training and p_drop > 0.0):
        N, L = qkv.shape[0], qkv.shape[1]
        if ATTN_TRAIN_BF16 and ops.gemm_precision() == "bf16" and qkv.is_cuda:
            return _AttentionTrainBf16.apply(qkv, mask, float(p_drop) if training else 0.0)
        return _AttentionTrain.apply(qkv, mask, float(p_drop) if training else 0.0)
    return ops.attention(qkv, mask)


# ------------------------------------------------------------------------------------------ fused training tower
TOWER_TRAIN_FUSED = True      # throughput mode, training: everything behind the input projection as 2 + 2 row kernels (tower_train.hip)


TOWER_TRAIN_MIXED = os.environ.get("DLDKD_TOWER_MIXED", "1") == "1"   # "mixed" precision: fp32-grade forward chain + the fused bf16 backward


def tower_train_mixed():
    """A tower built NOW runs as _TowerTrainMixed: precision "mixed", outside a backward pass."""
    return TOWER_TRAIN_MIXED and ops.precision_mode() == "mixed" and ops.gemm_precision() in ("fp32", "fp32x3", "fp32x2")


def tower_train_ok(is_cuda, L):
    return (TOWER_TRAIN_FUSED and (ops.gemm_precision() == "bf16" or tower_train_mixed()) and is_cuda and torch.is_grad_enabled()
            and 1 <= L <= 128)


def _bf16(shape, device):
    return torch.empty(shape, dtype=torch.bfloat16, device=device)


def _tt_pack(jobs, device, mask=None):
    """jobs = [(sources, mode)] -> the packed weights as views of one buffer; one kernel launch (dldkd_tower_train_pack).
    mode 4: sources = [w], any contiguous fp32 tensor with numel % 8 == 0 -> its plain bf16 cast (same shape).
    mask (n, L) fp32 contiguous: the sequence lengths (int32) come out of the same launch, appended to the result."""
    import ctypes
    L_ = _L()
    sizes = [(srcs[0].numel() * 2 + 15) // 16 * 16 if md == 4 else L_.dldkd_tower_train_pack_bytes(len(srcs)) for srcs, md in jobs]
    buf = torch.empty(sum(sizes), dtype=torch.uint8, device=device)
    outs, off = [], 0
    for sz in sizes:
        outs.append(buf[off:off + sz])
        off += sz
    n = len(jobs)
    src = (ctypes.c_void_p * (3 * n))()
    nsrc, mode, out = (ctypes.c_int * n)(), (ctypes.c_int * n)(), (ctypes.c_void_p * n)()
    for j, (srcs, md) in enumerate(jobs):
        if md == 4:
            w = srcs[0]
            if w.dtype != torch.float32 or not w.is_contiguous() or w.numel() % 8:
                raise native.NativeError("tower_train: the cast job needs a contiguous fp32 tensor of 8 k elements")
            src[3 * j] = w.data_ptr()
            nsrc[j], mode[j], out[j] = w.numel() // 8, 4, outs[j].data_ptr()
            outs[j] = outs[j][:w.numel() * 2].view(torch.bfloat16).view(w.shape)
            continue
        for c, w in enumerate(srcs):
            if w.dtype != torch.float32 or not w.is_contiguous() or tuple(w.shape) != (HIDDEN, HIDDEN):
                raise native.NativeError("tower_train: weights must be contiguous fp32 (384, 384)")
            src[3 * j + c] = w.data_ptr()
        nsrc[j], mode[j], out[j] = len(srcs), md, outs[j].data_ptr()
    if mask is not None:
        lens = torch.empty(mask.shape[0], dtype=torch.int32, device=device)
        native.check(L_.dldkd_tower_train_prepare(src, nsrc, mode, out, n, _p(mask), mask.shape[0], mask.shape[1], _p(lens), _s()),
                     "tower_train_prepare")
        return outs + [lens]
    native.check(L_.dldkd_tower_train_pack(src, nsrc, mode, out, n, _s()), "tower_train_pack")
    return outs


# One launch per tower and step for every weight operand: model._tower_fused calls tower_prepack() in front of the input projection;
# _InProjTrain.forward (the bf16 weight of its GEMM) and _TowerTrain.forward (the fragment packs) take what was prepared for THEIR
# weights (matched by storage and parameter epoch) and fall back to their own launches otherwise.
TOWER_PREPACK = os.environ.get("DLDKD_TOWER_PREPACK", "1") == "1"
_PREPACKED = {}
# _TowerTrain.backward -> _InProjTrain.backward: the bf16 copy of the gradient it returns (by the fp32 tensor's address; taken at once)
_DY16 = {}
# the LayerNorm parameter gradients of the fused towers: column sums inside the two backward row kernels (True: a third of their time)
# or beside the split-K reduce of the weight gradients, from bf16 rows the kernels leave instead (False)
TOWER_LN_SUMS_IN_KERNEL = os.environ.get("DLDKD_TOWER_LN_SUMS", "0") == "1"


def _tt_jobs(wq, wk, wv, wd, wo):
    return [([wq, wk, wv], 0), ([wd], 0)] + ([([wo], 1), ([wo], 2)] if wo is not None else []) + [([wd], 3), ([wq, wk, wv], 2)]


def tower_prepack(w_in, wq, wk, wv, wd, wo, mask=None, mixed=None):
    """Prepares (bf16 cast of the input projection's weight w_in - None: not wanted, fragment packs of the tower's matrices) in one
    launch and leaves them for the two consumers; with the batch's mask ((n, L) fp32) the same launch counts the sequence lengths,
    which are returned (int32; None without a usable mask).  mixed = (w_in or None, bq, bk, bv): in "mixed" precision also the
    two-plane forms of the forward weights (one more launch: split2_jobs)."""
    jobs = _tt_jobs(wq, wk, wv, wd, wo)
    cast = w_in is not None and w_in.is_contiguous() and w_in.dtype == torch.float32 and w_in.numel() % 8 == 0
    lens = None
    if mask is not None and not (mask.is_cuda and mask.dtype == torch.float32 and mask.is_contiguous() and mask.dim() == 2):
        mask = None
    outs = _tt_pack(jobs + ([([w_in], 4)] if cast else []), wq.device, mask=mask)
    if mask is not None:
        lens = outs.pop()
    _PREPACKED.clear()
    ep = ops.param_epoch()
    if cast:
        _PREPACKED["w16"] = (w_in.data_ptr(), ep, outs.pop())
    _PREPACKED["packs"] = (tuple(w.data_ptr() for w in (wq, wk, wv, wd) + ((wo,) if wo is not None else ())), ep, outs)
    if mixed is not None and tower_train_mixed() and MIXED_PLANES:
        # "mixed" precision: the tower's forward weights as two bf16 planes each (+ the three attention biases gathered), ONE launch
        w_in_m, bq, bk, bv = mixed
        jobs = ([(w_in_m, "split")] if w_in_m is not None else []) + [((wq, wk, wv), "split"), ((bq, bk, bv), "copy"), (wd, "split")] + \
               ([(wo, "split")] if wo is not None else [])
        res = split2_jobs(jobs, wq.device)
        if w_in_m is not None:
            _PREPACKED["w_in_planes"] = (w_in_m.data_ptr(), ep, res.pop(0))
        _PREPACKED["mixed_planes"] = (tuple(w.data_ptr() for w in (wq, wk, wv, wd) + ((wo,) if wo is not None else ())), ep, res)
    return lens


def _take_prepacked(kind, key):
    ent = _PREPACKED.pop(kind, None)
    if ent is not None and ent[0] == key and ent[1] == ops.param_epoch():
        return ent[2]
    return None


class _TowerTrain(Function):
    """One encoder tower behind its input projection, training, throughput mode (reference model_components.py:277-284, 398-450 and
    model.py:219): position LayerNorm + dropout -> q | k | v (one row kernel), attention (bf16 in / out), dense + dropout + residual +
    LayerNorm [+ out mapping] (one row kernel); the backward pass mirrors it with two row kernels around the attention backward
    kernel, weight gradients as GEMMs over the saved bf16 rows.  Dropout masks are recomputed from their Philox slots (the slots are
    drawn in the order of the unfused layers: the masks are the unfused path's, bit for bit)."""

    @staticmethod
    def forward(ctx, y0, pos, g1, b1, wq, bq, wk, bk, wv, bv, wd, bd, g2, b2, wo, bo, mask, lens, flags, p_in, p_attn, p_hid,
                relu_mask):
        N, L, _ = y0.shape
        M, dev = N * L, y0.device
        video = wo is not None
        if flags is None:
            lens = None          # the attention kernels skip the 32-row tiles past a sequence only where the row kernels skip them too
        packs = _take_prepacked("packs", tuple(w.data_ptr() for w in (wq, wk, wv, wd) + ((wo,) if video else ())))
        if packs is None:
            packs = _tt_pack(_tt_jobs(wq, wk, wv, wd, wo), dev)
        if video:
            pk_qkv, pk_d, pk_o, pk_ot, pk_dt, pk_qkvt = packs
        else:
            (pk_qkv, pk_d, pk_dt, pk_qkvt), pk_o, pk_ot = packs, None, None
        slot = lambda p, n: _philox_slot(dev, n) if p > 0.0 else (0, 0, None)      # noqa: E731
        sa, sb, sc = slot(p_in, M * HIDDEN), slot(p_attn, N * HEADS * L * L), slot(p_hid, M * HIDDEN)
        h1d, qkv, ctxl, xh2 = _bf16((M, HIDDEN), dev), _bf16((M, 3 * HIDDEN), dev), _bf16((M, HIDDEN), dev), _bf16((M, HIDDEN), dev)
        xh1 = _bf16((M, HIDDEN), dev)
        relu_bits = torch.empty(M * 48, dtype=torch.uint8, device=dev)       # [y0 > 0], one bit per element (tower_train.hip f1 / b1)
        stats = torch.empty(2, M, dtype=torch.float32, device=dev)
        rstd2 = torch.empty(M, dtype=torch.float32, device=dev)
        h2 = _bf16((M, HIDDEN), dev) if video else torch.empty(N, L, HIDDEN, dtype=torch.float32, device=dev)
        out = torch.empty(N, L, HIDDEN, dtype=torch.float32, device=dev) if video else h2
        L_ = _L()
        native.check(L_.dldkd_tower_train_f1(_p(y0), _p(pos), L, _p(g1), _p(b1), ops.LN_EPS, float(p_in), sa[0], sa[1], sa[2], _p(pk_qkv),
                                             _p(bq), _p(bk), _p(bv), _p(flags), M, _p(h1d), _p(xh1), _p(stats), _p(qkv), _p(relu_bits), _s()), "tower_train_f1")
        native.check(L_.dldkd_attention_train_fwd_bf16io(_p(qkv), _p(mask), _p(lens), _p(ctxl), N, L, float(p_attn), sb[0], sb[1], sb[2],
                                                         _s()), "attention_train_fwd_bf16io")
        native.check(L_.dldkd_tower_train_f3(_p(ctxl), _p(h1d), _p(pk_d), _p(bd), float(p_hid), sc[0], sc[1], sc[2], _p(g2), _p(b2),
                                             ops.LN_EPS, _p(pk_o), _p(bo), _p(flags), M, _p(xh2), _p(rstd2), _p(h2) if video else None,
                                             None if video else _p(h2), _p(out) if video else None, _s()), "tower_train_f3")
        ctx.save_for_backward(xh1, pos, g1, g2, mask, lens, flags, h1d, stats, qkv, ctxl, xh2, rstd2, h2 if video else None,
                              pk_ot, pk_dt, pk_qkvt, relu_bits)
        ctx.shape = (N, L)
        keep = _philox_step.dev if _philox_step is not None else None
        ctx.cfg = (video, float(p_in), float(p_attn), float(p_hid), sa, sb, sc, bool(relu_mask), keep)
        return out

    @staticmethod
    @ops.in_backward
    def backward(ctx, dout):
        xh1, pos, g1, g2, mask, lens, flags, h1d, stats, qkv, ctxl, xh2, rstd2, h2, pk_ot, pk_dt, pk_qkvt, relu_bits = ctx.saved_tensors
        video, p_in, p_attn, p_hid, sa, sb, sc, relu_mask, _keep = ctx.cfg
        N, L = ctx.shape
        M, dev = N * L, xh1.device
        dout = _f32(dout)
        L_ = _L()
        lnp = _zeros((4, HIDDEN), dev)                           # dgamma2, dbeta2, dgamma1, dbeta1 (atomics)
        ddo, dctx, dres, dqkv = _bf16((M, HIDDEN), dev), _bf16((M, HIDDEN), dev), _bf16((M, HIDDEN), dev), _bf16((M, 3 * HIDDEN), dev)
        sums = TOWER_LN_SUMS_IN_KERNEL
        # bf16 rows for the GEMMs / the finishing launch: dout under the out mapping (its weight gradient), the gradients of the two
        # LayerNorms' outputs (their parameter gradients), dy0 (the input projection's weight gradient)
        dg16 = _bf16((M, HIDDEN), dev) if video else None
        dh2_16 = _bf16((M, HIDDEN), dev) if (video and not sums) else None
        dz16 = None if sums else _bf16((M, HIDDEN), dev)
        dy16 = _bf16((M, HIDDEN), dev)
        native.check(L_.dldkd_tower_train_b3(_p(dout), _p(pk_ot), _p(xh2), _p(rstd2), _p(g2), p_hid, sc[0], sc[1], sc[2], _p(pk_dt),
                                             _p(flags), M, _p(ddo), _p(dctx), _p(dres), _p(lnp[0]) if sums else None,
                                             _p(lnp[1]) if sums else None, _p(dg16), _p(dh2_16), _s()), "tower_train_b3")
        native.check(L_.dldkd_attention_train_bwd_bf16io(_p(qkv), _p(mask), _p(lens), _p(dctx), _p(dqkv), N, L, p_attn, sb[0], sb[1],
                                                         sb[2], _s()), "attention_train_bwd_bf16io")
        need_pos = ctx.needs_input_grad[1]
        dy0 = torch.empty(N, L, HIDDEN, dtype=torch.float32, device=dev)
        dx1 = torch.empty(N, L * HIDDEN, dtype=torch.float32, device=dev) if need_pos else None
        native.check(L_.dldkd_tower_train_b1(_p(dqkv), _p(dres), _p(pk_qkvt), _p(xh1), _p(relu_bits), _p(stats), _p(g1), p_in, sa[0], sa[1],
                                             sa[2], _p(flags), M, int(relu_mask), _p(dy0), _p(dx1), _p(lnp[2]) if sums else None,
                                             _p(lnp[3]) if sums else None, _p(dz16), _p(dy16), _s()), "tower_train_b1")
        if len(_DY16) > 8:
            _DY16.clear()
        _DY16[dy0.data_ptr()] = dy16

        # weight + bias gradients: ONE split-K product over the saved bf16 rows, blocks [Wo |] Wd | Wq | Wk | Wv (gemm_bf16.hip)
        import ctypes
        blocks = ([(dg16, HIDDEN, 0, 1, h2)] if video else []) + [(ddo, HIDDEN, 0, 1, ctxl)] + [(dqkv, 3 * HIDDEN, c * HIDDEN, 1, h1d) for c in range(3)]
        nb = len(blocks)
        dW = torch.empty(nb * HIDDEN, HIDDEN, dtype=torch.float32, device=dev)
        dB = _zeros((nb * HIDDEN,), dev)
        ws_bytes = L_.dldkd_tower_train_dw_workspace_bytes(nb, M)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
        hA, hB = (ctypes.c_void_p * nb)(), (ctypes.c_void_p * nb)()
        hl, hc, h16 = (ctypes.c_int * nb)(), (ctypes.c_int * nb)(), (ctypes.c_int * nb)()
        for i, (a_, lda, col, a16, b_) in enumerate(blocks):
            hA[i], hB[i], hl[i], hc[i], h16[i] = a_.data_ptr(), b_.data_ptr(), lda, col, a16
        dpos = None
        if need_pos:                                         # gradient of the whole table: rows >= L stay zero (arena); summed over the
            dpos = _zeros((pos.shape[0], HIDDEN), dev)       # sequences inside the launch that reduces the split-K planes
        if not sums:                                         # ... which then sums the LayerNorm parameter gradients as well
            dh2 = dh2_16 if video else dout
            native.check(L_.dldkd_tower_train_dw_ln(hA, hl, hc, h16, hB, nb, M, _p(dW), _p(dB), _p(ws), ws_bytes, _p(flags), _p(dx1), _p(dpos),
                                                    dx1.shape[0] if need_pos else 0, L * HIDDEN, _p(dz16), _p(xh1), _p(dh2), 1 if video else 0,
                                                    _p(xh2), _p(lnp), _s()), "tower_train_dw_ln")
        elif need_pos:
            native.check(L_.dldkd_tower_train_dw_pos(hA, hl, hc, h16, hB, nb, M, _p(dW), _p(dB), _p(ws), ws_bytes, _p(flags), _p(dx1), _p(dpos),
                                                     dx1.shape[0], L * HIDDEN, _s()), "tower_train_dw_pos")
        else:
            native.check(L_.dldkd_tower_train_dw(hA, hl, hc, h16, hB, nb, M, _p(dW), _p(dB), _p(ws), ws_bytes, _p(flags), _s()), "tower_train_dw")
        H = HIDDEN
        o = 1 if video else 0
        dwo, dbo = (dW[:H], dB[:H]) if video else (None, None)
        dwd, dbd = dW[o * H:(o + 1) * H], dB[o * H:(o + 1) * H]
        dwq, dwk, dwv = (dW[(o + 1 + c) * H:(o + 2 + c) * H] for c in range(3))
        dbq, dbk, dbv = (dB[(o + 1 + c) * H:(o + 2 + c) * H] for c in range(3))
        return (dy0, dpos, lnp[2], lnp[3], dwq, dbq, dwk, dbk, dwv, dbv, dwd, dbd, lnp[0], lnp[1], dwo, dbo,
                None, None, None, None, None, None, None)


class _TowerTrainMixed(Function):
    """_TowerTrain in "mixed" precision (ops.set_gemm_precision("mixed")): the FORWARD pass is the fp32-grade kernel chain of the
    parity mode - LayerNorm + dropout, q | k | v as one three-plane GEMM, the exact fp32 attention, dense, dropout, residual
    LayerNorm [, out mapping] (model_components.py:277-284, 398-450; model.py:219) - so the tower's output, hence every loss value, is
    the parity mode's (north_star: 1e-4).  One launch (dldkd_tower_train_emit) then writes, from the fp32 intermediates, the bf16 rows
    _TowerTrain's fused backward kernels read, and the BACKWARD pass IS _TowerTrain.backward: bf16 products on exact activations.
    The dropout masks are the fused path's: same Philox slots, drawn in the same order, same element indexing."""

    @staticmethod
    def forward(ctx, y0, pos, g1, b1, wq, bq, wk, bk, wv, bv, wd, bd, g2, b2, wo, bo, mask, lens, flags, p_in, p_attn, p_hid,
                relu_mask):
        N, L, _ = y0.shape
        M, dev = N * L, y0.device
        video = wo is not None
        if flags is None:
            lens = None
        packs = _take_prepacked("packs", tuple(w.data_ptr() for w in (wq, wk, wv, wd) + ((wo,) if video else ())))
        if packs is None:
            packs = _tt_pack(_tt_jobs(wq, wk, wv, wd, wo), dev)
        if video:
            _, _, _, pk_ot, pk_dt, pk_qkvt = packs
        else:
            (_, _, pk_dt, pk_qkvt), pk_ot = packs, None
        slot = lambda p, n: _philox_slot(dev, n) if p > 0.0 else (0, 0, None)      # noqa: E731
        sa, sb, sc = slot(p_in, M * HIDDEN), slot(p_attn, N * HEADS * L * L), slot(p_hid, M * HIDDEN)
        L_ = _L()
        f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)   # noqa: E731
        y2 = y0.reshape(M, HIDDEN)
        planes = mixed_planes_ok(M, HIDDEN, HIDDEN)
        bf = lambda *shape: torch.empty(*shape, dtype=torch.bfloat16, device=dev)   # noqa: E731
        wqkv_p = bqkv = wd_p = wo_p = None
        if planes:
            mp = _take_prepacked("mixed_planes", tuple(w.data_ptr() for w in (wq, wk, wv, wd) + ((wo,) if video else ())))
            if mp is None:
                mp = split2_jobs([((wq, wk, wv), "split"), ((bq, bk, bv), "copy"), (wd, "split")] + ([(wo, "split")] if video else []), dev)
            wqkv_p, bqkv, wd_p = mp[0], mp[1], mp[2]
            wo_p = mp[3] if video else None
        # (1) h1 = dropout(LayerNorm(y0 + pos)), statistics kept [planes: also as the two-plane operand of q | k | v]
        h1, stats = f32(M, HIDDEN), f32(2, M)
        h1p = bf(2, M, HIDDEN) if planes else None
        native.check(L_.dldkd_layernorm_ex_f32(_p(y2), _p(pos), L, _p(g1), _p(b1), _p(h1), None, _p(h1p), None, _p(stats), M, HIDDEN, ops.LN_EPS,
                                               float(p_in), sa[0], sa[1], sa[2], None, None, _p(flags), _s()), "layernorm_ex")
        # (2) q | k | v: one fp32-grade GEMM (padding groups not multiplied)
        if planes:
            qkv = gemm_planes(h1p, wqkv_p, bqkv, M, 3 * HIDDEN, HIDDEN, row_flags=flags)
        else:
            qkv = ops.linear(h1, torch.cat([wq, wk, wv], 0), torch.cat([bq, bk, bv], 0), row_flags=flags)
        # (3) the exact fp32 attention (probabilities not written: the bf16 backward kernel recomputes them)
        ctxf = f32(N, L, HIDDEN)
        native.check(L_.dldkd_attention_train_fwd_f32(_p(qkv), _p(mask), None, _p(ctxf), N, L, float(p_attn), sb[0], sb[1], sb[2], _s()),
                     "attention_train_fwd")
        # (4) dense -> dropout -> + h1 -> LayerNorm [-> out mapping]
        ctxp = None
        if planes:
            (ctxp,) = split2_jobs([(ctxf.reshape(M, HIDDEN), "split")], dev)
            d = gemm_planes(ctxp, wd_p, bd, M, HIDDEN, HIDDEN, row_flags=flags)
        else:
            d = ops.linear(ctxf.reshape(M, HIDDEN), wd, bd, row_flags=flags)
        if p_hid > 0.0:
            dd, keep = torch.empty_like(d), torch.empty(d.shape, dtype=torch.uint8, device=dev)
            native.check(L_.dldkd_dropout_fwd_f32(_p(d), _p(dd), _p(keep), d.numel(), float(p_hid), sc[0], sc[1], sc[2], _s()), "dropout_fwd")
            del keep
        else:
            dd = d
        stats2 = f32(2, M)
        h2 = None if (planes and video) else f32(M, HIDDEN)               # (video towers on planes: the out mapping reads the planes only)
        h2p = bf(2, M, HIDDEN) if (planes and video) else None
        native.check(L_.dldkd_layernorm_ex_f32(_p(dd), _p(h1), 0, _p(g2), _p(b2), _p(h2), None, _p(h2p), None, _p(stats2), M, HIDDEN, ops.LN_EPS, 0.0,
                                               0, 0, None, None, None, _p(flags), _s()), "layernorm_ex")
        if video:
            out = (gemm_planes(h2p, wo_p, bo, M, HIDDEN, HIDDEN, row_flags=flags) if planes else ops.linear(h2, wo, bo, row_flags=flags)).view(N, L, HIDDEN)
        else:
            out = h2.view(N, L, HIDDEN)
        # (5) what the fused bf16 backward reads (on planes: h1d / ctx / h2 as bf16 are plane 0 of the GEMM operands - not written again)
        xh1, qkv16, xh2 = _bf16((M, HIDDEN), dev), _bf16((M, 3 * HIDDEN), dev), _bf16((M, HIDDEN), dev)
        h1d = h1p[0] if planes else _bf16((M, HIDDEN), dev)
        ctxl = ctxp[0] if planes else _bf16((M, HIDDEN), dev)
        rstd2 = f32(M)
        relu_bits = torch.empty(M * 48, dtype=torch.uint8, device=dev)
        h2_16 = (h2p[0] if planes else _bf16((M, HIDDEN), dev)) if video else None
        emit_h2 = video and not planes
        native.check(L_.dldkd_tower_train_emit(_p(y2), _p(pos), L, _p(stats), _p(h1), _p(qkv), None if planes else _p(ctxf), _p(dd), _p(stats2),
                                               _p(h2) if emit_h2 else None, _p(flags), M, _p(xh1), _p(relu_bits), None if planes else _p(h1d), _p(qkv16),
                                               None if planes else _p(ctxl), _p(xh2), _p(rstd2), _p(h2_16) if emit_h2 else None, _s()), "tower_train_emit")
        ctx.save_for_backward(xh1, pos, g1, g2, mask, lens, flags, h1d, stats, qkv16, ctxl, xh2, rstd2, h2_16, pk_ot, pk_dt, pk_qkvt, relu_bits)
        ctx.shape = (N, L)
        keep_alive = _philox_step.dev if _philox_step is not None else None
        ctx.cfg = (video, float(p_in), float(p_attn), float(p_hid), sa, sb, sc, bool(relu_mask), keep_alive)
        return out

    backward = _TowerTrain.backward


def tower_train(y0, pos, g1, b1, qkv_layers, dense, g2, b2, out_linear, mask, lens, flags, p_in, p_attn, p_hid, training, relu_mask=True):
    """See _TowerTrain.  qkv_layers = (query, key, value) nn.Linear, dense / out_linear nn.Linear (out_linear None: query towers,
    the result is the LayerNorm output h2).  Dropout rates apply when `training`."""
    q, k, v = qkv_layers
    pz = (lambda p: float(p) if training else 0.0)               # noqa: E731
    wo, bo = (out_linear.weight, out_linear.bias) if out_linear is not None else (None, None)
    fn = _TowerTrainMixed if tower_train_mixed() else _TowerTrain
    return fn.apply(_f32(y0), _f32(pos), g1, b1, q.weight, q.bias, k.weight, k.bias, v.weight, v.bias, dense.weight, dense.bias,
                    g2, b2, wo, bo, _f32(mask), lens, flags, pz(p_in), pz(p_attn), pz(p_hid), bool(relu_mask))


# ------------------------------------------------------------------------------------------ modular pooling
class _ModPool(Function):
    @staticmethod
    def forward(ctx, h, mask, w):
        out, attn = ops.modpool(h, mask, w, want_attn=True)
        ctx.save_for_backward(h, mask, w, attn)
        return out

    @staticmethod
    @ops.in_backward
    def backward(ctx, dout):
        h, mask, w, attn = ctx.saved_tensors
        dout = _f32(dout)
        dh = torch.empty_like(h)
        dw = _zeros(tuple(w.shape), w.device)
        native.check(_L().dldkd_modpool_bwd_f32(_p(h), _p(mask), _p(w), _p(attn), _p(dout), _p(dh), _p(dw), h.shape[0],
                                                h.shape[1], _s()), "modpool_bwd")
        return dh, None, dw


def modpool(h, mask, w):
    h, mask, w = _f32(h), _f32(mask), _f32(w)
    if _needs_grad(h, w):
        return _ModPool.apply(h, mask, w)
    return ops.modpool(h, mask, w)


# ------------------------------------------------------------------------------------------ scoring (training form)
class _Normalize(Function):
    @staticmethod
    def forward(ctx, x):
        D = x.shape[-1]
        y = torch.empty_like(x)
        inv = torch.empty(x.numel() // D, dtype=torch.float32, device=x.device)
        native.check(_L().dldkd_normalize_rows_fwd_f32(_p(x), _p(y), _p(inv), x.numel() // D, D, _s()), "normalize_rows_fwd")
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    @ops.in_backward
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        D = y.shape[-1]
        dy = _f32(dy)
        dx = torch.empty_like(y)
        native.check(_L().dldkd_normalize_rows_bwd_f32(_p(y), _p(inv), _p(dy), _p(dx), y.numel() // D, D, _s()),
                     "normalize_rows_bwd")
        return dx


def normalize(x):
    return _Normalize.apply(_f32(x))


class _ClipScores(Function):
    """S[q, v, l] = <q_q, g_{v,l}> as one fp32-grade GEMM: (Nq, D) x (Nv*L, D)^T (model.py:321,344)."""

    @staticmethod
    def forward(ctx, q, g):
        Nv, L, D = g.shape
        ctx.save_for_backward(q, g)
        return ops.linear(q, g.reshape(Nv * L, D)).view(q.shape[0], Nv, L)

    @staticmethod
    @ops.in_backward
    def backward(ctx, dS):
        q, g = ctx.saved_tensors
        Nv, L, D = g.shape
        Nq = q.shape[0]
        dS2 = _f32(dS).reshape(Nq, Nv * L)
        dq = ops.gemm(dS2, g.reshape(Nv * L, D), False, True, Nq, D, Nv * L) if ctx.needs_input_grad[0] else None
        dg = ops.gemm(dS2, q, True, True, Nv * L, D, Nq).view(Nv, L, D) if ctx.needs_input_grad[1] else None
        return dq, dg


def clip_scores(q, g):
    return _ClipScores.apply(_f32(q), _f32(g))


class _ClipPool(Function):
    """mask_logits + max over clips.  Returns (pooled (Nq, Nv), masked clip scores (Nq, Nv, L))."""

    @staticmethod
    def forward(ctx, S, lens):
        Nq, Nv, L = S.shape
        Sm = S.clone()
        pooled = torch.empty(Nq, Nv, dtype=torch.float32, device=S.device)
        arg = torch.empty(Nq, Nv, dtype=torch.int32, device=S.device)
        native.check(_L().dldkd_clip_pool_fwd_f32(_p(Sm), _p(lens), _p(pooled), _p(arg), Nq, Nv, L, _s()), "clip_pool_fwd")
        ctx.save_for_backward(arg, lens)
        ctx.shape = (Nq, Nv, L)
        ctx.mark_non_differentiable(arg)
        return pooled, Sm, arg

    @staticmethod
    @ops.in_backward
    def backward(ctx, dpooled, dSm, _darg):
        arg, lens = ctx.saved_tensors
        Nq, Nv, L = ctx.shape
        dS = _f32(dSm).clone() if dSm is not None else torch.zeros(Nq, Nv, L, dtype=torch.float32, device=arg.device)
        if dpooled is not None:
            native.check(_L().dldkd_clip_pool_bwd_f32(_p(_f32(dpooled)), _p(arg), _p(lens), _p(dS), Nq, Nv, L, _s()),
                         "clip_pool_bwd")
        return dS, None


def clip_pool(S, lens):
    return _ClipPool.apply(_f32(S), lens)


SIMPOOL_TRAIN_BF16_OPERANDS = os.environ.get("DLDKD_SIMPOOL_BF16_OPERANDS", "1") == "1"


class _SimPoolTrain(Function):
    """Fused training simpool (simpool_train.hip): (pooled cosine, pooled raw, positive-column clip cosines) of one
    (query set, gallery) pair from one MFMA GEMM with a pooling epilogue; backward = two gather kernels."""

    @staticmethod
    def forward(ctx, q, g, lens, labels, want_clip):
        Nq, D = q.shape
        Nv, L, _ = g.shape
        dev = q.device
        rq = torch.empty(Nq, dtype=torch.float32, device=dev)
        rg = torch.empty(Nv * L, dtype=torch.float32, device=dev)
        pc = torch.empty(Nq, Nv, dtype=torch.float32, device=dev)
        pr = torch.empty(Nq, Nv, dtype=torch.float32, device=dev)
        ac = torch.empty(Nq, Nv, dtype=torch.int32, device=dev)
        ar = torch.empty(Nq, Nv, dtype=torch.int32, device=dev)
        clip = torch.empty(Nq, L, dtype=torch.float32, device=dev) if want_clip else None
        if SIMPOOL_TRAIN_BF16_OPERANDS and ops.gemm_precision() == "bf16" and D % 64 == 0 and L <= 128:
            # throughput mode: the norm pass (which reads every row anyway) leaves the two operands as bf16 rows and the pooled GEMM
            # takes its tiles from those by LDS-DMA (gemm_bf16_nt16_pool_kernel); the backward gathers keep reading the fp32 rows
            q16 = torch.empty(Nq, D, dtype=torch.bfloat16, device=dev)
            g16 = torch.empty(Nv * L, D, dtype=torch.bfloat16, device=dev)
            native.check(_L().dldkd_row_invnorm2_cast_f32(_p(q), _p(rq), _p(q16), Nq, _p(g), _p(rg), _p(g16), Nv * L, D, _s()),
                         "row_invnorm2_cast")
            native.check(_L().dldkd_simpool_train_fwd_bf16in(_p(q16), _p(g16), _p(rq), _p(rg), _p(lens), _p(labels), Nq, Nv, L, D, _p(pc),
                                                             _p(pr), _p(ac), _p(ar), _p(clip), _s()), "simpool_train_fwd_bf16in")
        elif MIXED_PLANES and ops.gemm_precision() == "fp32x2" and D % 64 == 0 and L <= 128:
            # "mixed" precision: both operands as two bf16 planes (written by the norm pass, which reads every row anyway), the pooled
            # product in three K-long segments of the LDS-DMA kernel: the numbers of the in-kernel-split two-plane kernel
            qp = torch.empty(2, Nq, D, dtype=torch.bfloat16, device=dev)
            gp = torch.empty(2, Nv * L, D, dtype=torch.bfloat16, device=dev)
            native.check(_L().dldkd_row_invnorm2_planes_f32(_p(q), _p(rq), _p(qp), Nq, _p(g), _p(rg), _p(gp), Nv * L, D, _s()),
                         "row_invnorm2_planes")
            native.check(_L().dldkd_simpool_train_fwd_planes(_p(qp), _p(gp), _p(rq), _p(rg), _p(lens), _p(labels), Nq, Nv, L, D, _p(pc), _p(pr),
                                                             _p(ac), _p(ar), _p(clip), _s()), "simpool_train_fwd_planes")
        else:
            native.check(_L().dldkd_row_invnorm2_f32(_p(q), _p(rq), Nq, _p(g), _p(rg), Nv * L, D, _s()), "row_invnorm2")
            native.check(_L().dldkd_simpool_train_fwd_f32(ops._PREC_ID[ops.gemm_precision()], _p(q), _p(g), _p(rq), _p(rg), _p(lens),
                                                          _p(labels), Nq, Nv, L, D, _p(pc), _p(pr), _p(ac), _p(ar), _p(clip), _s()),
                         "simpool_train_fwd")
        ctx.save_for_backward(q, g, rq, rg, lens, labels, ac, ar, pc, clip)
        ctx.mark_non_differentiable(ac, ar)
        ctx.set_materialize_grads(False)       # unused outputs (the arg-max planes, a clip row nobody reads) arrive as None in backward
                                               # instead of freshly filled zero tensors (2 int + up to 2 float fill kernels per call)
        if clip is None:
            return pc, pr, ac, ar
        return pc, pr, ac, ar, clip

    @staticmethod
    @ops.in_backward
    def backward(ctx, d_cos, d_raw, _dac, _dar, d_clip=None):
        q, g, rq, rg, lens, labels, ac, ar, pc, clip = ctx.saved_tensors
        Nq, D = q.shape
        Nv, L, _ = g.shape
        dq = torch.empty_like(q) if ctx.needs_input_grad[0] else None
        dg = torch.empty_like(g) if ctx.needs_input_grad[1] else None
        f = lambda t: None if t is None else _f32(t)            # noqa: E731
        native.check(_L().dldkd_simpool_train_bwd_f32(_p(q), _p(g), _p(rq), _p(rg), _p(lens), _p(labels), _p(ac), _p(ar), _p(pc),
                                                      _p(clip), _p(f(d_cos)), _p(f(d_raw)), _p(f(d_clip)), Nq, Nv, L, D, _p(dq),
                                                      _p(dg), _s()), "simpool_train_bwd")
        return dq, dg, None, None, None


def simpool_train_ok():
    """The pooled GEMM exists for the parity ("fp32" = three bf16 planes) and the throughput ("bf16") precisions."""
    return ops.gemm_precision() in ("fp32", "fp32x3", "fp32x2", "bf16")


def simpool_train(q, g, lens, labels, want_clip):
    """(pooled_cos (Nq, Nv), pooled_raw (Nq, Nv), clip_pos (Nq, L) or None) - see _SimPoolTrain."""
    out = _SimPoolTrain.apply(_f32(q), _f32(g), lens, labels, bool(want_clip))
    return out[0], out[1], (out[4] if want_clip else None)


# ------------------------------------------------------------------------------------------ losses
def _gscalar(g):
    """The upstream gradient of a scalar loss as a contiguous fp32 device scalar: the loss kernels read it on the device
    (float(g) would be one host synchronisation per loss per step)."""
    return g if (g.dtype == torch.float32 and g.is_contiguous()) else g.float().contiguous()


class _SumScalars(Function):
    """loss = t0 + t1 + ... of scalar terms, left to right (the reference's order, model.py:157-160), as ONE launch; every term's
    gradient is the upstream gradient itself."""

    @staticmethod
    def forward(ctx, *ts):
        import ctypes
        ctx.n = len(ts)
        out = torch.empty((), dtype=torch.float32, device=ts[0].device)
        ptrs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        native.check(_L().dldkd_sum_scalars_f32(ptrs, len(ts), _p(out), _s()), "sum_scalars")
        return out

    @staticmethod
    @ops.in_backward
    def backward(ctx, g):
        return (g,) * ctx.n


def sum_scalars(*ts):
    """Sum of 0-dim fp32 GPU tensors in the given order; falls back to torch adds for anything else."""
    if 1 < len(ts) <= 8 and all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.numel() == 1 for t in ts):
        return _SumScalars.apply(*[t.reshape(()) if t.dim() else t for t in ts])
    out = ts[0]
    for t in ts[1:]:
        out = out + t
    return out


def _sum(x):
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    native.check(_L().dldkd_sum_f32(_p(x), x.numel(), _p(out), _s()), "sum")
    return out.reshape(())


class _KLFrame(Function):
    @staticmethod
    def forward(ctx, Sp, St, labels, lens, temp):
        Nq, Nv, L = Sp.shape if Sp.dim() == 3 else (Sp.shape[0], 0, Sp.shape[1])      # (Nq, L): the positive column only
        out = torch.empty(Nq, dtype=torch.float32, device=Sp.device)
        native.check(_L().dldkd_kl_frame_f32(_p(Sp), _p(St), _p(labels), _p(lens), temp, Nq, Nv, L, _p(out), None, None, _s()),
                     "kl_frame")
        ctx.save_for_backward(Sp, St, labels, lens)
        ctx.temp = temp
        return _sum(out)

    @staticmethod
    @ops.in_backward
    def backward(ctx, g):
        Sp, St, labels, lens = ctx.saved_tensors
        Nq, Nv, L = Sp.shape if Sp.dim() == 3 else (Sp.shape[0], 0, Sp.shape[1])
        dSp = _zeros(tuple(Sp.shape), Sp.device)
        native.check(_L().dldkd_kl_frame_f32(_p(Sp), _p(St), _p(labels), _p(lens), ctx.temp, Nq, Nv, L, None, _p(dSp),
                                             _p(_gscalar(g)), _s()), "kl_frame")
        return dSp, None, None, None, None


def kl_frame(Sp, St, labels, lens, temp=0.2):
    """sum_q KL(softmax(St[q, label_q, :len]/temp) || softmax(Sp[...]/temp)) (model.py:183-197)."""
    return _KLFrame.apply(_f32(Sp), _f32(St).detach(), labels, lens, float(temp))


class _NCE(Function):
    @staticmethod
    def forward(ctx, S, T, labels, cq, cv, hardQ, hardV, beta, eps, t_is_s):
        Nq, Nv = S.shape
        terms = torch.empty(Nq + Nv, dtype=torch.float32, device=S.device)
        native.check(_L().dldkd_nce_f32(_p(S), _p(T), _p(labels), _p(cq), _p(cv), hardQ, hardV, beta, eps, Nq, Nv, _p(terms),
                                        None, None, None, _s()), "nce")
        ctx.save_for_backward(S, T, labels, cq, cv)
        ctx.cfg = (hardQ, hardV, beta, eps, t_is_s)
        return _sum(terms)

    @staticmethod
    @ops.in_backward
    def backward(ctx, g):
        S, T, labels, cq, cv = ctx.saved_tensors
        hardQ, hardV, beta, eps, t_is_s = ctx.cfg
        Nq, Nv = S.shape
        dS = torch.empty_like(S)
        dT = torch.empty_like(S) if (T is not None and t_is_s) else None
        native.check(_L().dldkd_nce_f32(_p(S), _p(T), _p(labels), _p(cq), _p(cv), hardQ, hardV, beta, eps, Nq, Nv, None, _p(dS),
                                        _p(dT), _p(_gscalar(g)), _s()), "nce")
        if dT is not None:          # the exploration branch's soft labels are its own scores (model.py:149-150)
            _axpy(dS, dT)
        return dS, None, None, None, None, None, None, None, None, None


_COEF_CACHE = {}


def _part_coefs(n, hard_n, w_hard, w_soft, device):
    """c[:hard_n] = w_hard, c[hard_n:] = w_soft as a device vector.  The vectors depend on (batch shape, alpha) only, so they are
    built once (two fills on the device - no pageable host-to-device copy: that blocks the host and cannot be graph-captured)
    and kept: the 12 fills per step were 4.7 us each in the replayed step.  A vector first needed while a capture is open is
    built inside it and not kept (its memory belongs to the graph's pool)."""
    key = (int(n), int(hard_n), float(w_hard), float(w_soft), str(device))
    c = _COEF_CACHE.get(key)
    if c is not None:
        return c
    c = torch.full((n,), float(w_soft), dtype=torch.float32, device=device)
    if hard_n > 0:
        c[:hard_n] = float(w_hard)
    if torch.device(device).type == "cuda" and not torch.cuda.is_current_stream_capturing():
        _COEF_CACHE[key] = c              # never evicted: captured graphs read these vectors by address (a few KB per (shape, alpha))
    return c


def nce_soft(labels, S, T, alpha, beta):
    """clip_nce_soft.forward (model_components.py:126-199).  T may be S itself (gradient then also flows
    through the soft targets) or a tensor without gradient (teacher scores)."""
    S = _f32(S)
    Nq, Nv = S.shape
    hardQ, hardV = math.floor(alpha * Nq), math.floor(alpha * Nv)
    softQ, softV = Nq - hardQ, Nv - hardV
    use_hard = hardQ != 0 and hardV != 0
    use_soft = softQ != 0 and softV != 0
    cq = _part_coefs(Nq, hardQ, alpha / hardQ if use_hard else 0.0, (1 - alpha) / softQ if use_soft else 0.0, S.device)
    cv = _part_coefs(Nv, hardV, alpha / hardV if use_hard else 0.0, (1 - alpha) / softV if use_soft else 0.0, S.device)
    t_is_s = T is S
    Tt = S.detach() if t_is_s else _f32(T).detach()
    return _NCE.apply(S, Tt, labels, cq, cv, hardQ, hardV, float(beta), 1e-12, t_is_s)


def nce_hard(labels, S):
    """clip_nce.forward (model_components.py:216-234)."""
    S = _f32(S)
    Nq, Nv = S.shape
    cq = _part_coefs(Nq, 0, 0.0, 1.0 / Nq, S.device)
    cv = _part_coefs(Nv, 0, 0.0, 1.0 / Nv, S.device)
    return _NCE.apply(S, None, labels, cq, cv, Nq, Nv, 0.0, 0.0, False)


BRANCH_LOSS_FUSED = True      # a branch's triplet + InfoNCE + KL terms and their gradients as three launches (losses_f32.hip)


_UNIT_GRADS = {}


def unit_grad(device):
    """The constant 1.0 (0-dim fp32) a caller passes as the upstream gradient of loss terms it only sums (train.GraphedTrainStep's
    branch graphs: grad_outputs of torch.autograd.grad).  _BranchLoss.backward recognises it by its storage and skips the scaling
    launch - the gradients were computed for an upstream gradient of 1.  Made once per device, outside any capture."""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else 0))
    t = _UNIT_GRADS.get(key)
    if t is None:
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("unit_grad: first use on a device must be outside a stream capture")
        t = _UNIT_GRADS[key] = torch.ones((), dtype=torch.float32, device=dev)
    return t


def _is_unit(g):
    t = _UNIT_GRADS.get((g.device.type, g.device.index if g.device.index is not None else 0))
    return t is not None and g.data_ptr() == t.data_ptr() and g.numel() == 1


class ScheduleWords:
    """The scalars an epoch's schedule moves (train.epoch_schedules: alpha, belta, the KD weight; train.py:66-113) as DEVICE state a
    captured step reads by address: the InfoNCE coefficient vectors cq (Nq) / cv (Nv) and, per loss weight w_kl, five words
    {hardQ, hardV, bits(belta), bits(w_kl), nq_valid} (dldkd_branch_losses_f32 `sched`; nq_valid: the batch's real queries when the
    query axis is padded to a bucket).  train.GraphedTrainStep makes one per captured step,
    installs it around the capture (`schedule_words`) and calls update() before every replay: a graph captured in epoch 0 serves
    every epoch (by value the scalars were part of the graph's key: a capture per epoch, and none after max_captures).
    update() enqueues a few fills on the current stream when - and only when - the values changed."""

    def __init__(self, nq, nv, soft, device, store=None):
        """store: an int32 device tensor of nq + nv + 10 words to live in - [cq | cv | 2 x 5 words] (train.GraphedTrainStep: a
        range of the step's staged words, written on the host with write() and uploaded with them); None: own tensors, rewritten
        on the device by update()."""
        self.nq, self.nv, self.soft, self.device = int(nq), int(nv), bool(soft), torch.device(device)
        self.store = store
        if store is None:
            self.cq = torch.zeros(self.nq, dtype=torch.float32, device=self.device)
            self.cv = torch.zeros(self.nv, dtype=torch.float32, device=self.device)
        else:
            if store.dtype != torch.int32 or store.numel() != self.words_needed(nq, nv):
                raise ValueError("ScheduleWords: store must be int32 of nq + nv + 10 words")
            self.cq, self.cv = self.split(store, self.nq, self.nv)[:2]
        self.words = {}                   # w_kl factor (host float, e.g. kl_intra_weight or 0.0) -> (5,) int32 device words
        self.state = None
        self.used = 0                     # branch_losses calls that took their scalars from here

    MAX_FACTORS = 2

    @classmethod
    def words_needed(cls, nq, nv):
        return int(nq) + int(nv) + 5 * cls.MAX_FACTORS

    @classmethod
    def split(cls, t, nq, nv):
        """(cq, cv, [5-word slices]) views of an int32 tensor laid out like `store` (device words or their pinned host slot)."""
        w = t[nq + nv:]
        return t[:nq].view(torch.float32), t[nq:nq + nv].view(torch.float32), [w[5 * i:5 * i + 5] for i in range(cls.MAX_FACTORS)]

    def words_for(self, factor):
        """The device words of a branch whose KL weight is `factor` x the epoch's KD weight (made while the step is captured)."""
        f = float(factor)
        if f not in self.words:
            if self.store is not None:
                if len(self.words) >= self.MAX_FACTORS:
                    raise RuntimeError("ScheduleWords: more KL factors than word slots")
                self.words[f] = self.split(self.store, self.nq, self.nv)[2][len(self.words)]
            else:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("ScheduleWords: a branch's words must exist before the capture (prepare())")
                self.words[f] = torch.zeros(5, dtype=torch.int32, device=self.device)
            self.state = None
        return self.words[f]

    @staticmethod
    def coefs(n_q, n_v, alpha, soft):
        """(hardQ, hardV, (wq_hard, wq_soft), (wv_hard, wv_soft)) of clip_nce_soft / clip_nce (model_components.py:126-234)."""
        if not soft:
            return n_q, n_v, (0.0, 1.0 / n_q), (0.0, 1.0 / n_v)
        hq, hv = math.floor(alpha * n_q), math.floor(alpha * n_v)
        sq, sv = n_q - hq, n_v - hv
        use_hard, use_soft = hq != 0 and hv != 0, sq != 0 and sv != 0
        return (hq, hv, (alpha / hq if use_hard else 0.0, (1 - alpha) / sq if use_soft else 0.0),
                (alpha / hv if use_hard else 0.0, (1 - alpha) / sv if use_soft else 0.0))

    def update(self, alpha, beta, weight, nq_valid=None):
        """nq_valid: the batch's number of real queries (<= nq: the query axis is padded to a bucket, dldkd_branch_losses_f32)."""
        nqv = self.nq if nq_valid is None else int(nq_valid)
        if not 0 < nqv <= self.nq:
            raise ValueError(f"ScheduleWords: {nqv} valid queries of {self.nq}")
        state = (float(alpha), float(beta), float(weight), nqv, tuple(self.words))
        if state == self.state:
            return False
        self._fill(self.cq, self.cv, [self.words[f] for f in self.words], alpha, beta, weight, nqv)
        self.state = state
        return True

    def write(self, host_words, alpha, beta, weight, nq_valid=None):
        """The same values into a HOST tensor laid out like `store` (the step's pinned staging slot): uploaded with the slot."""
        nqv = self.nq if nq_valid is None else int(nq_valid)
        if not 0 < nqv <= self.nq:
            raise ValueError(f"ScheduleWords: {nqv} valid queries of {self.nq}")
        cq, cv, w = self.split(host_words, self.nq, self.nv)
        self._fill(cq, cv, w[:len(self.words)], alpha, beta, weight, nqv)

    def _fill(self, cq, cv, words, alpha, beta, weight, nqv):
        hq, hv, (qh, qs), (vh, vs) = self.coefs(nqv, self.nv, float(alpha), self.soft)
        for c, h, wh, ws in ((cq, hq, qh, qs), (cv, hv, vh, vs)):
            c.fill_(float(ws))
            if self.soft and h > 0:
                c[:h].fill_(float(wh))
        b = float(beta) if self.soft else 0.0
        for f, w in zip(self.words, words):
            w[0:1].fill_(int(hq)); w[1:2].fill_(int(hv))
            w.view(torch.float32)[2:3].fill_(b)
            w.view(torch.float32)[3:4].fill_(float(f * float(weight)))
            w[4:5].fill_(nqv)


_SCHED = None


@contextlib.contextmanager
def schedule_words(sw):
    """branch_losses calls inside read the schedule's scalars from `sw` (device) instead of baking the host values in."""
    global _SCHED
    prev, _SCHED = _SCHED, sw
    try:
        yield sw
    finally:
        _SCHED = prev


class _BranchLoss(Function):
    """The loss terms of one branch (model.py:137-155) from its pooled scores, values and gradients in one pass: the gradients are
    computed with the values (for an upstream gradient of 1) and scaled by the actual upstream gradients in the backward pass."""

    @staticmethod
    def forward(ctx, C, S, T, clip_p, clip_t, labels, lens, r_t2v, r_v2t, cq, cv, cfg, sched=None, nq_valid=0):
        hard, hardQ, hardV, fold_t, margin, beta, eps, temp, w_nce, w_kl = cfg
        nq, nv = C.shape
        dev = C.device
        Lc = clip_p.shape[1] if clip_p is not None else 0
        terms = torch.empty(2 * (nq + nv) + nq, dtype=torch.float32, device=dev)
        out = torch.empty(3, dtype=torch.float32, device=dev)
        dC = _zeros((nq, nv), dev)
        dS = torch.empty(nq, nv, dtype=torch.float32, device=dev)
        dclip = _zeros((nq, Lc), dev) if clip_p is not None else None
        native.check(_L().dldkd_branch_losses_f32(_p(C), _p(S), _p(T), _p(clip_p), _p(clip_t), _p(labels), _p(lens), _p(r_t2v), _p(r_v2t),
                                                  _p(cq), _p(cv), nq, nv, Lc, int(hard), int(hardQ), int(hardV), int(fold_t), float(margin),
                                                  float(beta), float(eps), float(temp), float(w_nce), float(w_kl), _p(terms), _p(dC), _p(dS),
                                                  _p(dclip), _p(out), int(nq_valid), _p(sched), _s()), "branch_losses")
        ctx.save_for_backward(dC, dS, dclip)
        ctx.set_materialize_grads(False)       # a term nobody uses has no upstream gradient (None, handled in backward): no zero fill
        return out[0], out[1], out[2]

    @staticmethod
    @ops.in_backward
    def backward(ctx, g_trip, g_nce, g_kl):
        dC, dS, dclip = ctx.saved_tensors
        if (g_trip is not None and g_nce is not None and _is_unit(g_trip) and _is_unit(g_nce)
                and (dclip is None or (g_kl is not None and _is_unit(g_kl)))):
            return dC, dS, None, dclip, None, None, None, None, None, None, None, None, None, None      # scaling by exactly 1: nothing to launch
        # The saved gradients are scaled IN PLACE and handed out as they are (no second copy of two (Nq, Nv) matrices per branch): a
        # second backward pass through this node (retain_graph=True) would scale them twice.  Once only, loudly.
        if getattr(ctx, "_scaled", False):
            raise RuntimeError("_BranchLoss: second backward through a node whose saved gradients were scaled in place; rebuild the "
                               "forward pass instead of retain_graph=True (or sum the terms with unit upstream gradients)")
        ctx._scaled = True
        one = None
        def gs(g):                                             # an unused term has no upstream gradient: its gradients are zero
            nonlocal one
            if g is not None:
                return _gscalar(g)
            if one is None:
                one = _zeros((1,), dC.device)
            return one
        native.check(_L().dldkd_branch_losses_scale_f32(_p(dC), _p(dS), dC.numel(), _p(dclip), 0 if dclip is None else dclip.numel(),
                                                        _p(gs(g_trip)), _p(gs(g_nce)), _p(gs(g_kl)), _s()), "branch_losses_scale")
        return dC, dS, None, dclip, None, None, None, None, None, None, None, None, None, None


def branch_losses(C, S, T, clip_p, clip_t, labels, lens, r_t2v, r_v2t, hard, margin, soft, alpha, beta, w_nce, w_kl, fold_t, kd_factor=None,
                  nq_valid=None):
    """(triplet, w_nce * InfoNCE, w_kl * KL) of one branch.  soft: clip_nce_soft with soft-label scores T (fold_t: T is S itself,
    exploration branch), else clip_nce.  clip_p None: no KL term (the third value is 0).  kd_factor: w_kl = kd_factor x the epoch's KD
    weight (what a ScheduleWords needs to follow the schedule; None: w_kl is a constant of the run).  nq_valid: rows [nq_valid, Nq) of
    the score matrices belong to padding queries (None: none) - terms, normalisers and coefficients are those of nq_valid queries, the
    padding rows get zero gradients."""
    C, S = _f32(C), _f32(S)
    Nq, Nv = S.shape
    nqv = Nq if nq_valid is None else int(nq_valid)
    if not 0 < nqv <= Nq:
        raise native.NativeError(f"branch_losses: {nqv} valid queries of {Nq}")
    sw = _SCHED
    if sw is not None and kd_factor is not None and (sw.nq, sw.nv, sw.soft) == (Nq, Nv, bool(soft)) and S.is_cuda:
        # the schedule's scalars as device state (ScheduleWords): this launch reads hardQ / hardV / belta / w_kl and the coefficient
        # vectors by address; the by-value arguments are the current epoch's (what the words hold right now) and are ignored
        hardQ, hardV, _, _ = sw.coefs(nqv, Nv, float(alpha), bool(soft))
        eps, Tt = (1e-12, None if fold_t else _f32(T).detach()) if soft else (0.0, None)
        if not soft:
            fold_t, beta = False, 0.0
        cfg = (bool(hard), hardQ, hardV, bool(fold_t), float(margin), float(beta), eps, 0.2, float(w_nce), float(w_kl))
        sw.used += 1
        return _BranchLoss.apply(C, S, Tt, None if clip_p is None else _f32(clip_p), None if clip_t is None else _f32(clip_t).detach(), labels,
                                 lens, r_t2v, r_v2t, sw.cq, sw.cv, cfg, sw.words_for(kd_factor), nqv)
    if soft:
        hardQ, hardV = math.floor(alpha * nqv), math.floor(alpha * Nv)
        softQ, softV = nqv - hardQ, Nv - hardV
        use_hard = hardQ != 0 and hardV != 0
        use_soft = softQ != 0 and softV != 0
        cq = _part_coefs(Nq, hardQ, alpha / hardQ if use_hard else 0.0, (1 - alpha) / softQ if use_soft else 0.0, S.device)
        cv = _part_coefs(Nv, hardV, alpha / hardV if use_hard else 0.0, (1 - alpha) / softV if use_soft else 0.0, S.device)
        eps, Tt = 1e-12, (None if fold_t else _f32(T).detach())
    else:
        hardQ, hardV, eps, Tt, fold_t, beta = nqv, Nv, 0.0, None, False, 0.0
        cq = _part_coefs(Nq, 0, 0.0, 1.0 / nqv, S.device)
        cv = _part_coefs(Nv, 0, 0.0, 1.0 / Nv, S.device)
    cfg = (bool(hard), hardQ, hardV, bool(fold_t), float(margin), float(beta), eps, 0.2, float(w_nce), float(w_kl))
    return _BranchLoss.apply(C, S, Tt, None if clip_p is None else _f32(clip_p), None if clip_t is None else _f32(clip_t).detach(), labels, lens,
                             r_t2v, r_v2t, cq, cv, cfg, None, nqv)


class _Triplet(Function):
    @staticmethod
    def forward(ctx, C, labels, r_t2v, r_v2t, hard, margin):
        Nq, Nv = C.shape
        terms = torch.empty(Nq + Nv, dtype=torch.float32, device=C.device)
        native.check(_L().dldkd_triplet_f32(_p(C), _p(labels), _p(r_t2v), _p(r_v2t), int(hard), margin, Nq, Nv, _p(terms),
                                            None, None, _s()), "triplet")
        ctx.save_for_backward(C, labels, r_t2v, r_v2t)
        ctx.cfg = (hard, margin)
        return _sum(terms)

    @staticmethod
    @ops.in_backward
    def backward(ctx, g):
        C, labels, r_t2v, r_v2t = ctx.saved_tensors
        hard, margin = ctx.cfg
        Nq, Nv = C.shape
        dC = _zeros(tuple(C.shape), C.device)
        native.check(_L().dldkd_triplet_f32(_p(C), _p(labels), _p(r_t2v), _p(r_v2t), int(hard), margin, Nq, Nv, None, _p(dC),
                                            _p(_gscalar(g)), _s()), "triplet")
        return dC, None, None, None, None, None


def triplet(C, labels, r_t2v, r_v2t, hard, margin):
    """get_clip_triplet_loss (model.py:353-387) with the reference's random draws passed in."""
    return _Triplet.apply(_f32(C), labels, r_t2v, r_v2t, bool(hard), float(margin))
