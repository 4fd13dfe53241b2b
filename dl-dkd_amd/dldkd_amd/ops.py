"""Thin tensor-level wrappers over the C ABI (one function per kernel).  No math happens here."""
import os

import torch

from . import native

HIDDEN = 384
LN_EPS = 1e-5   # nn.LayerNorm default (model_components.py:274,301,443)


_PRECISION = "fp32"

# Packed / folded bf16 weight caches (FoldedInProj, PackedLinear) are keyed on (data_ptr, tensor._version) of their source
# parameters AND on this counter.  The fused optimizer updates parameters through raw device pointers, which changes
# neither - so BertAdam.step() (and anything else that writes parameter memory behind torch's back) bumps the counter and
# every cache repacks on its next use.  Without it, validation after a training step ran on the initial q|k|v / dense /
# out_mapping / input-projection weights.
_PARAM_EPOCH = 0


def bump_param_epoch():
    global _PARAM_EPOCH
    _PARAM_EPOCH += 1


def param_epoch():
    return _PARAM_EPOCH


_MIXED = False      # "mixed": parity-grade forward, bf16 backward
_BWD_DEPTH = 0
# the forward GEMMs of "mixed": "fp32x2" = two bf16 planes per operand, three MFMAs per product (gemm_f32x3.hip NPL = 2: ~2^-16 per
# product; the seven losses stay inside 1e-4 with a wide margin, measured) or "fp32" = the parity mode's three planes
MIXED_FORWARD = os.environ.get("DLDKD_MIXED_FORWARD", "fp32x2")


def set_gemm_precision(precision):
    """"fp32" (default, parity grade): fp32-grade products from three bf16 planes per operand on the bf16 matrix
    cores (gemm_f32x3.hip, ~2^-24 relative error per product, 1.5x the fp32-input MFMA); "fp32_exact": the true
    fp32-input MFMA (gemm_f32.hip).  "bf16": the GEMMs of `linear` / `gemm` (and the
    batched attention products in functional.py) run on bf16 MFMA with fp32 accumulation - the throughput mode of
    the training step (BASELINE.json configs[2]).
    "mixed" (round 6): the FORWARD pass as "fp32" - only the forward products (input projection, q|k|v, dense, out mapping,
    attention, simpool) determine the seven loss values (method/model.py:100-163), so they meet north_star's 1e-4 - and every
    BACKWARD pass (autograd Functions of functional.py, which read the precision when they run) as "bf16": the gradients carry
    bf16-product noise like the throughput mode's, on the exact forward activations."""
    global _PRECISION, _MIXED
    if precision not in ("fp32", "fp32_exact", "fp32x3", "fp32x2", "bf16", "mixed"):
        raise ValueError(precision)
    _MIXED = precision == "mixed"
    _PRECISION = MIXED_FORWARD if _MIXED else precision


def precision_mode():
    """What set_gemm_precision was given ("mixed" included); gemm_precision() is what a GEMM issued NOW runs in."""
    return "mixed" if _MIXED else _PRECISION


class backward_scope:
    """Entered by every autograd backward of functional.py: in "mixed" mode the GEMMs issued inside run as "bf16"."""

    def __enter__(self):
        global _PRECISION, _BWD_DEPTH
        if _MIXED:
            if _BWD_DEPTH == 0:
                _PRECISION = "bf16"
            _BWD_DEPTH += 1
            self.on = True
        else:
            self.on = False
        return self

    def __exit__(self, *exc):
        global _PRECISION, _BWD_DEPTH
        if self.on:
            _BWD_DEPTH -= 1
            if _BWD_DEPTH == 0 and _MIXED:
                _PRECISION = MIXED_FORWARD
        return False


def in_backward(fn):
    """Decorator of an autograd Function's backward (under @staticmethod): runs it inside backward_scope."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        with backward_scope():
            return fn(*a, **k)
    return wrapped


def gemm_precision():
    return _PRECISION


# ---- row groups of a padded batch (training, throughput mode).  The input projection of a video tower flags the 32-row groups of
# its (n, L, .) batch that hold valid clips (functional._InProjTrain); while that tower is being built the flags are current here
# and every row-wise kernel of the tower - linears, LayerNorms, their backward passes - skips the other groups: rows of the padding
# are never read, their values are act(bias) / zeros (finite don't-cares: padded KEYS are masked out of attention, padded clips out
# of every loss), their gradients exact zeros.
_ROW_GROUPS = None       # (uint8 flags (M / 32), M)


def set_row_groups(flags, M):
    global _ROW_GROUPS
    _ROW_GROUPS = None if flags is None else (flags, int(M))


def row_groups(M):
    """The current tower's group flags when a tensor of M rows is one row per (item, position) of its padded batch."""
    rg = _ROW_GROUPS
    return rg[0] if (rg is not None and rg[1] == int(M) and _PRECISION in ("bf16", "fp32", "fp32x3", "fp32x2")) else None


GEMM_NT_DMA = True       # throughput mode: forward-layout GEMMs with >= 1024 rows on the LDS-DMA staged kernel


def _gemm_fn(L):
    if _PRECISION == "bf16":
        return L.dldkd_gemm_bf16
    if _PRECISION in ("fp32", "fp32x3", "fp32x2"):   # fp32-grade on the bf16 matrix cores (three-plane split), the default; the two-plane
        return L.dldkd_gemm_f32x3                     # form exists for the forward layout only (linear below): other layouts take three
    return L.dldkd_gemm_f32                     # "fp32_exact": the true fp32-input MFMA


_PREC_ID = {"fp32_exact": 0, "fp32": 1, "fp32x3": 1, "bf16": 2, "fp32x2": 3}     # DLDKD_GEMM_F32 / _F32X3 / _BF16 / _F32X2


def _gemm_workspace(L, M, N, K, a_kmajor, b_kmajor, device, precision=None):
    """Split-K scratch for one GEMM call, from torch's caching allocator (stream-ordered, so it is safe under graph capture
    and with several streams); None when the shape never splits.  The library itself never allocates."""
    pid = _PREC_ID[precision or _PRECISION]
    nbytes = L.dldkd_gemm_workspace_bytes(1 if pid == 3 else pid, M, N, K, int(a_kmajor), int(b_kmajor))   # (fp32x2: other layouts run three planes)
    if nbytes == 0:
        return None, 0
    return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


def _chk(t, name):
    if t is not None and (not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous()):
        raise native.NativeError(f"{name}: need a contiguous fp32 GPU tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")
    return t


def linear(x, weight, bias=None, relu=False, row_flags=None):
    """y = act(x @ weight.T + bias); x (..., K), weight (N, K).  row_flags (uint8 per 32 rows, throughput mode): groups flagged 0
    are rows of the padding - not read; their output rows are act(bias)."""
    L = native.lib()
    K = x.shape[-1]
    x2 = _chk(x.reshape(-1, K), "linear.x")
    _chk(weight, "linear.weight"); _chk(bias, "linear.bias")
    M, N = x2.shape[0], weight.shape[0]
    if weight.shape[1] != K:
        raise native.NativeError(f"linear: x has {K} features, weight expects {weight.shape[1]}")
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    if GEMM_NT_DMA and _PRECISION == "bf16" and M >= 1024 and L.dldkd_gemm_bf16_nt_ok(M, N, K, K, K):
        # throughput mode, many rows: operand tiles by LDS-DMA (gemm_bf16_dma.hip; bit-identical to dldkd_gemm_bf16)
        native.check(L.dldkd_gemm_bf16_nt(native.ptr(x2), native.ptr(weight), native.ptr(bias), native.ptr(y), M, N, K, K, K, N,
                                          int(relu), native.ptr(row_flags), native.stream()), "gemm_bf16_nt")
        return y.view(*x.shape[:-1], N)
    if _PRECISION == "fp32x2":
        native.check(L.dldkd_gemm_f32x2(native.ptr(x2), native.ptr(weight), native.ptr(bias), native.ptr(y), M, N, K, K, K, N, int(relu),
                                        native.ptr(row_flags), native.stream()), "gemm_f32x2")
        return y.view(*x.shape[:-1], N)
    if row_flags is not None and _PRECISION in ("fp32", "fp32x3"):
        native.check(L.dldkd_gemm_f32x3_flags(native.ptr(x2), native.ptr(weight), native.ptr(bias), native.ptr(y), M, N, K, K, K, N,
                                              0, 0, int(relu), None, 0, native.ptr(row_flags), native.stream()), "gemm_f32x3_flags")
        return y.view(*x.shape[:-1], N)
    fn = _gemm_fn(L)
    native.check(fn(native.ptr(x2), native.ptr(weight), native.ptr(bias), native.ptr(y), M, N, K, K, K, N,
                    0, 0, int(relu), None, 0, native.stream()), "gemm")          # the forward layout never splits K
    return y.view(*x.shape[:-1], N)


def gemm(a, b, a_kmajor, b_kmajor, M, N, K, row_flags=None):
    """C[M,N] = sum_k A(m,k) B(n,k) with explicit operand layouts (used by the backward passes).  row_flags (throughput mode): per
    32 rows of the ACTIVATION operand(s) - of A's rows in the dX layout, of the contraction index in the dW layout; groups
    flagged 0 are rows of the padding (zero rows of dy) and are skipped."""
    L = native.lib()
    _chk(a, "gemm.a"); _chk(b, "gemm.b")
    c = torch.empty(M, N, dtype=torch.float32, device=a.device)
    if row_flags is not None and _PRECISION == "bf16" and a_kmajor and b_kmajor:
        ws, ws_bytes = _gemm_workspace(L, M, N, K, True, True, a.device)
        native.check(L.dldkd_gemm_bf16_mixed(2, native.ptr(a), native.ptr(b), None, native.ptr(c), M, N, K, a.shape[-1], b.shape[-1], N, 0,
                                             native.ptr(ws), ws_bytes, native.ptr(row_flags), native.stream()), "gemm_bf16_mixed")
        return c
    if (GEMM_NT_DMA and _PRECISION == "bf16" and not a_kmajor and b_kmajor and M >= 1024 and b.dim() == 2 and b.shape[0] == K
            and L.dldkd_gemm_bf16_nt_ok(M, N, K, a.shape[-1], K)):
        # dX = dy . W with a weight of a few hundred rows: transpose W (K x N, < 2 MB) and the product has the forward layout
        bt = b.t().contiguous()
        native.check(L.dldkd_gemm_bf16_nt(native.ptr(a), native.ptr(bt), None, native.ptr(c), M, N, K, a.shape[-1], K, N, 0,
                                          native.ptr(row_flags), native.stream()), "gemm_bf16_nt")
        return c
    fn = _gemm_fn(L)
    ws, ws_bytes = _gemm_workspace(L, M, N, K, a_kmajor, b_kmajor, a.device)
    if row_flags is not None and _PRECISION in ("fp32", "fp32x3", "fp32x2") and (not a_kmajor or b_kmajor):
        native.check(L.dldkd_gemm_f32x3_flags(native.ptr(a), native.ptr(b), None, native.ptr(c), M, N, K, a.shape[-1], b.shape[-1], N,
                                              int(a_kmajor), int(b_kmajor), 0, native.ptr(ws), ws_bytes, native.ptr(row_flags),
                                              native.stream()), "gemm_f32x3_flags")
        return c
    native.check(fn(native.ptr(a), native.ptr(b), None, native.ptr(c), M, N, K, a.shape[-1], b.shape[-1], N,
                    int(a_kmajor), int(b_kmajor), 0, native.ptr(ws), ws_bytes, native.stream()), "gemm")
    return c


def layernorm(x, gamma, beta, add=None, add_mod=0):
    L = native.lib()
    D = x.shape[-1]
    x2 = _chk(x.reshape(-1, D), "layernorm.x")
    _chk(gamma, "layernorm.gamma"); _chk(beta, "layernorm.beta"); _chk(add, "layernorm.add")
    out = torch.empty_like(x2)
    native.check(L.dldkd_layernorm_f32(native.ptr(x2), native.ptr(add), int(add_mod), native.ptr(gamma), native.ptr(beta),
                                       native.ptr(out), x2.shape[0], D, LN_EPS, native.stream()), "layernorm")
    return out.view(x.shape)


def attention(qkv, mask):
    """qkv (N, L, 1152), mask (N, L) or None -> context (N, L, 384)."""
    L = native.lib()
    _chk(mask, "attention.mask")
    N, Lq = qkv.shape[0], qkv.shape[1]
    out = torch.empty(N, Lq, HIDDEN, dtype=torch.float32, device=qkv.device)
    if qkv.dtype == torch.bfloat16:              # produced by linear_rows(out_bf16=True) in throughput mode
        if not (qkv.is_cuda and qkv.is_contiguous()):
            raise native.NativeError("attention.qkv: bf16 input must be a contiguous GPU tensor")
        native.check(L.dldkd_attention_fwd_bf16(native.ptr(qkv), native.ptr(mask), native.ptr(out), N, Lq, 1, native.stream()),
                     "attention_fwd_bf16")
        return out
    _chk(qkv, "attention.qkv")
    if _PRECISION == "bf16":
        native.check(L.dldkd_attention_fwd_bf16(native.ptr(qkv), native.ptr(mask), native.ptr(out), N, Lq, 0, native.stream()),
                     "attention_fwd_bf16")
    else:
        native.check(L.dldkd_attention_fwd_f32(native.ptr(qkv), native.ptr(mask), native.ptr(out), N, Lq, native.stream()),
                     "attention_fwd")
    return out


def modpool(h, mask, w, want_attn=False):
    """h (N, L, 384), mask (N, L), w (384,) -> (N, 384) [, attn (N, L)]."""
    L = native.lib()
    _chk(h, "modpool.h"); _chk(mask, "modpool.mask"); _chk(w, "modpool.w")
    N, Lw = h.shape[0], h.shape[1]
    out = torch.empty(N, HIDDEN, dtype=torch.float32, device=h.device)
    attn = torch.empty(N, Lw, dtype=torch.float32, device=h.device) if want_attn else None
    native.check(L.dldkd_modpool_fwd_f32(native.ptr(h), native.ptr(mask), native.ptr(w), native.ptr(out), native.ptr(attn),
                                         N, Lw, native.stream()), "modpool_fwd")
    return (out, attn) if want_attn else out


# K4 kernel for the two-branch projection: "rows128" = in_proj_rows128_kernel (round 2; needs K % 64 == 0, else falls back),
# "full" = rows_linear_bf16_kernel<1, true> (round 1).  Same contract; results differ in fp32 summation order only.
INPROJ_KERNEL = "rows128"


class FoldedInProj:
    """LayerNorm-folded bf16 weights of 1-2 LinearLayer modules (one per branch), rebuilt when a parameter
    changes (tensor._version, data_ptr, or the optimizer's raw-pointer update: _PARAM_EPOCH)."""

    def __init__(self, layers, full_row=None):
        self.layers = layers
        self.key = None
        # two branches: the full-row kernels (weights in MFMA fragment order); full_row=False forces the column-tiled kernel
        self.full_row = (len(layers) == 2) if full_row is None else full_row

    def _params(self):
        ps = []
        for l in self.layers:
            ps += [l.LayerNorm.weight, l.LayerNorm.bias, l.net[1].weight, l.net[1].bias]
        return ps

    def get(self):
        ps = self._params()
        key = (_PARAM_EPOCH,) + tuple((p.data_ptr(), p._version) for p in ps)
        if key != self.key:
            L = native.lib()
            K = self.layers[0].net[1].weight.shape[1]
            nb = len(self.layers)
            dev = ps[0].device
            self.Wf = torch.empty(nb * HIDDEN * K * 2, dtype=torch.uint8, device=dev)
            self.cs = torch.empty(nb * HIDDEN, dtype=torch.float32, device=dev)
            self.bb = torch.empty(nb * HIDDEN, dtype=torch.float32, device=dev)
            import ctypes
            for b, l in enumerate(self.layers):
                lin = l.net[1]
                if self.full_row:
                    native.check(L.dldkd_fold_ln_linear_h16_frag(
                        native.ptr(lin.weight.detach().contiguous()), native.ptr(lin.bias.detach()),
                        native.ptr(l.LayerNorm.weight.detach()), native.ptr(l.LayerNorm.bias.detach()), HIDDEN, K, b * HIDDEN,
                        native.ptr(self.Wf), native.ptr(self.cs), native.ptr(self.bb), native.stream()), "fold_ln_linear_frag")
                    continue
                native.check(L.dldkd_fold_ln_linear_h16(
                    native.ptr(lin.weight.detach().contiguous()), native.ptr(lin.bias.detach()),
                    native.ptr(l.LayerNorm.weight.detach()), native.ptr(l.LayerNorm.bias.detach()), HIDDEN, K,
                    ctypes.c_void_p(self.Wf.data_ptr() + b * HIDDEN * K * 2), ctypes.c_void_p(self.cs.data_ptr() + 4 * b * HIDDEN),
                    ctypes.c_void_p(self.bb.data_ptr() + 4 * b * HIDDEN), native.stream()), "fold_ln_linear")
            self.key, self.K, self.nb = key, K, nb
        return self


class FoldedInProjX3:
    """Parity-grade counterpart of FoldedInProj for TWO LinearLayer modules (one per branch): gamma folded into the fp32
    weights, split into three bf16 planes in the fragment order of in_proj_rows128x3_kernel; rebuilt when a parameter changes."""

    def __init__(self, layers):
        if len(layers) != 2:
            raise ValueError("FoldedInProjX3 packs the two branches' input projections")
        self.layers = layers
        self.key = None

    def get(self):
        ps = []
        for l in self.layers:
            ps += [l.LayerNorm.weight, l.LayerNorm.bias, l.net[1].weight, l.net[1].bias]
        key = (_PARAM_EPOCH,) + tuple((p.data_ptr(), p._version) for p in ps)
        if key != self.key:
            L = native.lib()
            K = self.layers[0].net[1].weight.shape[1]
            dev = ps[0].device
            self.Wp = torch.empty(3 * 2 * HIDDEN * K * 2, dtype=torch.uint8, device=dev)
            self.bb = torch.empty(2 * HIDDEN, dtype=torch.float32, device=dev)
            for b, l in enumerate(self.layers):
                lin = l.net[1]
                native.check(L.dldkd_fold_ln_linear_planes(
                    native.ptr(lin.weight.detach().contiguous()), native.ptr(lin.bias.detach()), native.ptr(l.LayerNorm.weight.detach()),
                    native.ptr(l.LayerNorm.bias.detach()), HIDDEN, K, b * HIDDEN, native.ptr(self.Wp), native.ptr(self.bb),
                    native.stream()), "fold_ln_linear_planes")
            self.key, self.K = key, K
        return self


def in_proj_x3_ok(K):
    return bool(native.lib().dldkd_in_proj_f32x3_rows128_ok(int(K)))


def in_proj_x3(x, folded, relu=True):
    """x (..., K) fp32 -> [branch 0, branch 1] (..., 384) fp32: LayerNorm + Linear (+ ReLU) of both branches with fp32-grade
    products, one pass over x (after a row-statistics pass)."""
    L = native.lib()
    f = folded.get()
    K = x.shape[-1]
    if K != f.K:
        raise native.NativeError(f"in_proj_x3: x has {K} features, weights expect {f.K}")
    x2 = _chk(x.reshape(-1, K), "in_proj_x3.x")
    M = x2.shape[0]
    stats = torch.empty(2, M, dtype=torch.float32, device=x.device)
    native.check(L.dldkd_row_meanrstd_f32(native.ptr(x2), native.ptr(stats[0]), native.ptr(stats[1]), M, K, LN_EPS, native.stream()),
                 "row_meanrstd")
    ys = [torch.empty(M, HIDDEN, dtype=torch.float32, device=x.device) for _ in range(2)]
    native.check(L.dldkd_in_proj_f32x3_rows128(native.ptr(x2), native.ptr(stats[0]), native.ptr(stats[1]), native.ptr(f.Wp),
                                               native.ptr(f.bb), native.ptr(ys[0]), native.ptr(ys[1]), M, K, int(relu),
                                               native.stream()), "in_proj_f32x3_rows128")
    return [y.view(*x.shape[:-1], HIDDEN) for y in ys]


class PackedLinear:
    """bf16 MFMA-fragment-order copy of 1-3 nn.Linear modules with the same in_features and 384 outputs each (one
    q|k|v block, or a single dense layer) for `linear_rows`; rebuilt when a parameter changes (tensor._version)."""

    def __init__(self, linears):
        self.linears = list(linears)
        if not 1 <= len(self.linears) <= 3 or any(l.weight.shape[0] != HIDDEN for l in self.linears):
            raise native.NativeError("PackedLinear: 1-3 linears with 384 outputs each")
        self.key = None

    def get(self):
        ps = [t for l in self.linears for t in (l.weight, l.bias)]
        key = (_PARAM_EPOCH,) + tuple((t.data_ptr(), t._version) for t in ps)
        if key != self.key:
            L = native.lib()
            K = self.linears[0].weight.shape[1]
            dev = ps[0].device
            # launches: the first two linears share one N = 768 pass, a third gets its own N = 384 pass
            groups = [self.linears[:2]] + ([self.linears[2:]] if len(self.linears) == 3 else [])
            self.groups = []
            for g in groups:
                n_total = HIDDEN * len(g)
                wf = torch.empty(n_total * K * 2, dtype=torch.uint8, device=dev)
                bb = torch.empty(n_total, dtype=torch.float32, device=dev)
                for i, l in enumerate(g):
                    native.check(L.dldkd_pack_linear_h16_frag(native.ptr(l.weight.detach().contiguous()), native.ptr(l.bias.detach()),
                                                               HIDDEN, K, i * HIDDEN, n_total, native.ptr(wf), native.ptr(bb),
                                                               native.stream()), "pack_linear_frag")
                self.groups.append((wf, bb, n_total))
            self.key, self.K = key, K
        return self


class PackedLinearX3:
    """Three-bf16-plane copy (fragment order of in_proj_rows128x3_kernel) of 1-3 nn.Linear modules with the same in_features and
    384 outputs each, for `linear_rows_x3` (parity mode, inference); rebuilt when a parameter changes."""

    def __init__(self, linears):
        self.linears = list(linears)
        if not 1 <= len(self.linears) <= 3 or any(l.weight.shape[0] != HIDDEN for l in self.linears):
            raise native.NativeError("PackedLinearX3: 1-3 linears with 384 outputs each")
        self.key = None

    def get(self):
        ps = [t for l in self.linears for t in (l.weight, l.bias)]
        key = (_PARAM_EPOCH,) + tuple((t.data_ptr(), t._version) for t in ps)
        if key != self.key:
            L = native.lib()
            K = self.linears[0].weight.shape[1]
            dev = ps[0].device
            groups = [self.linears[:2]] + ([self.linears[2:]] if len(self.linears) == 3 else [])   # N = 768 pass (+ N = 384 pass)
            self.groups = []
            for g in groups:
                n_total = HIDDEN * len(g)
                wp = torch.empty(3 * n_total * K * 2, dtype=torch.uint8, device=dev)
                bb = torch.empty(n_total, dtype=torch.float32, device=dev)
                for i, l in enumerate(g):
                    native.check(L.dldkd_pack_linear_planes(native.ptr(l.weight.detach().contiguous()), native.ptr(l.bias.detach()), None,
                                                            None, HIDDEN, K, i * HIDDEN, n_total, native.ptr(wp), native.ptr(bb),
                                                            native.stream()), "pack_linear_planes")
                self.groups.append((wp, bb, n_total))
            self.key, self.K = key, K
        return self


def rows_x3_ok(x):
    """The fp32-grade full-row kernel serves inference in parity mode: no autograd, in_features a multiple of 32 in [64, 4096]."""
    return (_PRECISION in ("fp32", "fp32x3") and ROWS_X3 and not torch.is_grad_enabled() and in_proj_x3_ok(x.shape[-1])
            and x.is_cuda)


ROWS_X3 = True        # parity inference: 384/768-wide linears of the towers on in_proj_rows128x3_kernel instead of gemm_f32x3


def linear_rows_x3(x, packed, relu=False):
    """x (..., K) fp32 -> (..., 384 * n_linears) fp32: y = act(x W^T + b) for the packed linears side by side, fp32-grade
    products (three bf16 planes), batch-invariant."""
    import ctypes
    L = native.lib()
    f = packed.get()
    K = x.shape[-1]
    if K != f.K:
        raise native.NativeError(f"linear_rows_x3: x has {K} features, weights expect {f.K}")
    x2 = _chk(x.reshape(-1, K), "linear_rows_x3.x")
    M = x2.shape[0]
    n_out = HIDDEN * len(f.linears)
    y = torch.empty(M, n_out, dtype=torch.float32, device=x.device)
    col = 0
    for wp, bb, n_total in f.groups:
        y0 = ctypes.c_void_p(y.data_ptr() + 4 * col)
        y1 = ctypes.c_void_p(y.data_ptr() + 4 * (col + HIDDEN)) if n_total == 2 * HIDDEN else None
        native.check(L.dldkd_linear_f32x3_rows(native.ptr(x2), None, None, native.ptr(wp), native.ptr(bb), y0, y1, M, n_total, K,
                                               n_out, int(relu), native.stream()), "linear_f32x3_rows")
        col += n_total
    return y.view(*x.shape[:-1], n_out)


def rows_kernel_ok(x):
    """The full-row bf16 kernel serves inference in throughput mode: no autograd, in_features a multiple of 32."""
    return _PRECISION == "bf16" and not torch.is_grad_enabled() and x.shape[-1] % 32 == 0


def linear_rows(x, packed, relu=False, out_bf16=False):
    """x (..., K) fp32 -> (..., 384 * n_linears) fp32 (bf16 with out_bf16): y = act(x W^T + b) for the packed linears
    side by side."""
    import ctypes
    L = native.lib()
    f = packed.get()
    K = x.shape[-1]
    if K != f.K:
        raise native.NativeError(f"linear_rows: x has {K} features, weights expect {f.K}")
    x2 = _chk(x.reshape(-1, K), "linear_rows.x")
    M = x2.shape[0]
    n_out = HIDDEN * len(f.linears)
    y = torch.empty(M, n_out, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    esz = 2 if out_bf16 else 4
    col = 0
    for wf, bb, n_total in f.groups:
        y0 = ctypes.c_void_p(y.data_ptr() + esz * col)
        y1 = ctypes.c_void_p(y.data_ptr() + esz * (col + HIDDEN)) if n_total == 2 * HIDDEN else None
        native.check(L.dldkd_linear_rows_h16(native.ptr(x2), native.ptr(wf), native.ptr(bb), y0, y1, n_out, M, n_total, K, int(relu),
                                              int(out_bf16), native.stream()), "linear_rows_bf16")
        col += n_total
    return y.view(*x.shape[:-1], n_out)


def plan_row_groups(lens, L):
    """Host side of in_proj_h16(groups=...): the 32-row groups of a padded (n, L, K) batch that hold valid clips, as int32 first
    rows (video v, clips 32 t ..: v L + 32 t, t < ceil(len_v / 32)), padded to a multiple of 4 by repeating the last group."""
    import numpy as np
    lens = np.asarray(lens, dtype=np.int64)
    if L % 32:
        raise native.NativeError("plan_row_groups: L must be a multiple of 32")
    nt = np.minimum((lens + 31) // 32, L // 32)
    v = np.repeat(np.arange(len(lens), dtype=np.int64), nt)
    t = np.arange(int(nt.sum()), dtype=np.int64) - np.repeat(np.cumsum(nt) - nt, nt)
    g = v * L + 32 * t
    if len(g) == 0:
        return np.zeros(0, np.int32)
    if len(g) % 4:
        g = np.concatenate([g, np.full(4 - len(g) % 4, g[-1], np.int64)])
    return np.ascontiguousarray(g.astype(np.int32))


def in_proj_h16(x, folded, relu=True, groups=None):
    """x (..., K) fp32 -> list of per-branch (..., 384) fp32 outputs, one pass over x (K4).  groups (int32 GPU tensor from
    plan_row_groups, rows128 kernel only): only those 32-row groups are projected; the other output rows stay unwritten."""
    L = native.lib()
    f = folded.get()
    K = x.shape[-1]
    if K != f.K:
        raise native.NativeError(f"in_proj: x has {K} features, weights expect {f.K}")
    x2 = _chk(x.reshape(-1, K), "in_proj.x")
    M = x2.shape[0]
    ys = [torch.empty(M, HIDDEN, dtype=torch.float32, device=x.device) for _ in range(f.nb)]
    if groups is not None:
        if not (f.full_row and INPROJ_KERNEL == "rows128" and L.dldkd_in_proj_h16_rows128_ok(K)):
            raise native.NativeError("in_proj_h16: a row-group table needs the two-branch rows128 kernel")
        if groups.dtype != torch.int32 or not groups.is_cuda:
            raise native.NativeError("in_proj_h16: groups must be an int32 GPU tensor")
        native.check(L.dldkd_in_proj_h16_rows128_groups(native.ptr(x2), native.ptr(f.Wf), native.ptr(f.cs), native.ptr(f.bb),
                                                         native.ptr(ys[0]), native.ptr(ys[1]), M, K, LN_EPS, int(relu),
                                                         native.ptr(groups), groups.numel(), native.stream()), "in_proj_bf16_rows128_groups")
        return [y.view(*x.shape[:-1], HIDDEN) for y in ys]
    if f.full_row:
        rows128 = INPROJ_KERNEL == "rows128" and L.dldkd_in_proj_h16_rows128_ok(K)
        fn = L.dldkd_in_proj_h16_rows128 if rows128 else L.dldkd_in_proj_h16_full
        native.check(fn(native.ptr(x2), native.ptr(f.Wf), native.ptr(f.cs), native.ptr(f.bb), native.ptr(ys[0]),
                        native.ptr(ys[1]), M, K, LN_EPS, int(relu), native.stream()), "in_proj_bf16_" + INPROJ_KERNEL)
        return [y.view(*x.shape[:-1], HIDDEN) for y in ys]
    native.check(L.dldkd_in_proj_h16(native.ptr(x2), native.ptr(f.Wf), native.ptr(f.cs), native.ptr(f.bb), native.ptr(ys[0]),
                                      native.ptr(ys[1]) if f.nb == 2 else None, M, f.nb * HIDDEN, K, LN_EPS, int(relu),
                                      native.stream()), "in_proj_h16")
    return [y.view(*x.shape[:-1], HIDDEN) for y in ys]


# ------------------------------------------------------------------------------------ K4b: projection of resident 16-bit rows
# "h16" everywhere below = IEEE fp16, the operand format of the eval-path towers (csrc/common.hpp says why it is not bf16)
H16 = torch.float16


class ResidentRows:
    """Raw feature rows in their device-resident form: fp16 (rows, K) + fp32 LayerNorm statistics per row (mean, rstd), ragged
    (item i owns rows [start[i], start[i] + lens[i])).  Filled by appending padded fp32 batches (dldkd_rows_to_h16_stats)."""

    def __init__(self, K, device, capacity_rows=0):
        self.K, self.device = int(K), torch.device(device)
        self.rows = 0
        self.lens = []                                   # host: rows per item, in append order
        self._alloc(max(int(capacity_rows), 1))

    def _alloc(self, cap):
        cap = -(-cap // 128) * 128 + 128                 # slack: whole 128-row tiles
        xb = torch.empty(cap, self.K, dtype=H16, device=self.device)
        mean = torch.empty(cap, dtype=torch.float32, device=self.device)
        rstd = torch.empty(cap, dtype=torch.float32, device=self.device)
        if self.rows:
            xb[:self.rows] = self.xb[:self.rows]
            mean[:self.rows] = self.mean[:self.rows]
            rstd[:self.rows] = self.rstd[:self.rows]
        self.xb, self.mean, self.rstd, self.cap = xb, mean, rstd, cap

    def append(self, feat, lens_host):
        """feat (n, L, K) fp32 GPU (a padded batch), lens_host (n) ints: the valid rows of every item join the table."""
        import numpy as np
        lens_host = np.asarray(lens_host, dtype=np.int64)
        n, L, K = feat.shape
        if K != self.K or len(lens_host) != n or (lens_host > L).any() or (lens_host < 0).any():
            raise native.NativeError("ResidentRows.append: batch does not match the table")
        add = int(lens_host.sum())
        if self.rows + add > self.cap - 128:
            self._alloc(max(2 * self.cap, self.rows + add))
        start = self.rows + np.concatenate([[0], np.cumsum(lens_host)[:-1]]) if n else np.zeros(0, np.int64)
        meta = torch.from_numpy(np.concatenate([start.astype(np.int64), lens_host])).to(self.device)   # (2 n) int64: one upload
        lens_d = meta[n:].to(torch.int32)
        x = _chk(feat.reshape(n * L, K), "ResidentRows.append")
        native.check(native.lib().dldkd_rows_to_h16_stats(native.ptr(x), native.ptr(lens_d), native.ptr(meta), n, L, K, LN_EPS,
                                                           native.ptr(self.xb), native.ptr(self.mean), native.ptr(self.rstd),
                                                           native.stream()), "rows_to_h16_stats")
        self.lens.extend(int(v) for v in lens_host)
        self.rows += add

    def append_rows(self, rows, item_lens=()):
        """rows (R, K) fp32 GPU: R more rows join the table as they lie (ragged: no padding ever existed); item_lens: the lengths of
        the items those rows COMPLETE, in order (an item's rows may arrive in two calls).  Same kernel as append (one "item" of R rows)."""
        import numpy as np
        R, K = rows.shape
        if K != self.K:
            raise native.NativeError("ResidentRows.append_rows: rows do not match the table")
        if R:
            if self.rows + R > self.cap - 128:
                self._alloc(max(2 * self.cap, self.rows + R))
            meta = torch.from_numpy(np.asarray([self.rows, R], dtype=np.int64)).to(self.device)
            lens_d = meta[1:].to(torch.int32)
            x = _chk(rows, "ResidentRows.append_rows")
            native.check(native.lib().dldkd_rows_to_h16_stats(native.ptr(x), native.ptr(lens_d), native.ptr(meta), 1, R, K, LN_EPS,
                                                               native.ptr(self.xb), native.ptr(self.mean), native.ptr(self.rstd),
                                                               native.stream()), "rows_to_h16_stats")
            self.rows += R
        self.lens.extend(int(v) for v in item_lens)

    def clear(self):
        self.rows, self.lens = 0, []

    def nbytes(self):
        return self.rows * (self.K * 2 + 8)


def in_proj_rows_ok(K):
    return bool(native.lib().dldkd_in_proj_h16_rows128b_ok(int(K)))


# the resident gallery encode hands h0 from K4b to the fused tower as fp16 rows (DLDKD_H0_H16=0: fp32 rows, for A/B runs)
RESIDENT_H0_H16 = os.environ.get("DLDKD_H0_H16", "1") == "1"
# the resident gallery encode into a zero-filled (or reused) packed gallery leaves the rows the scorers never load unwritten
SKIP_ZERO_ROWS = os.environ.get("DLDKD_SKIP_ZERO_ROWS", "1") == "1"


def in_proj_resident(table, row_lo, row_hi, folded, relu=True, out=None, out_h16=False):
    """K4b: rows [row_lo, row_hi) of a ResidentRows table -> per-branch (row_hi - row_lo, 384) fp32 (both branches, one pass);
    out_h16: fp16 rows instead (what tower_seq's gallery mode reads with half the traffic)."""
    L = native.lib()
    f = folded.get()
    if not (f.full_row and f.nb == 2 and f.K == table.K and in_proj_rows_ok(table.K)):
        raise native.NativeError("in_proj_resident: needs the two-branch fragment-order weights and K % 64 == 0, K >= 256")
    M = int(row_hi - row_lo)
    if row_lo < 0 or row_hi > table.rows or M < 0:
        raise native.NativeError("in_proj_resident: row range outside the table")
    dt = H16 if out_h16 else torch.float32
    ys = out if out is not None else [torch.empty(M, HIDDEN, dtype=dt, device=table.device) for _ in range(2)]
    if any(y.shape[0] < M or y.shape[1] != HIDDEN or y.dtype != dt or not y.is_contiguous() for y in ys):
        raise native.NativeError("in_proj_resident: out tensors must be contiguous %s (>= rows, 384)" % dt)
    if M == 0:
        return ys
    fn = L.dldkd_in_proj_h16_rows128b_out16 if out_h16 else L.dldkd_in_proj_h16_rows128b
    native.check(fn(native.ptr(table.xb[row_lo:]), native.ptr(table.mean[row_lo:]), native.ptr(table.rstd[row_lo:]), native.ptr(f.Wf), native.ptr(f.cs), native.ptr(f.bb),
                    native.ptr(ys[0]), native.ptr(ys[1]), M, table.K, int(relu), None, 0, native.stream()), "in_proj_bf16_rows128b")
    return ys


# ---------------------------------------------------------------------------------------------- K5: fused per-sequence tower
TOWER_SEQ = True      # throughput-mode inference: everything behind the input projection as one kernel (tower_seq.hip)


class TowerPack:
    """One branch's tower behind the input projection (position LayerNorm, q | k | v, dense + LayerNorm, then the out mapping of
    a video tower or the modular pooling vector of a query tower) as the bf16 fragment blob of tower_seq_kernel; rebuilt when a
    parameter changes (like PackedLinear)."""

    def __init__(self, pos_embed, encoder, out_linear=None, mod_linear=None):
        if (out_linear is None) == (mod_linear is None):
            raise native.NativeError("TowerPack: a video tower has an out mapping, a query tower a modular vector mapping")
        self.pos_embed, self.encoder, self.out_linear, self.mod_linear = pos_embed, encoder, out_linear, mod_linear
        self.key = None

    def _params(self):
        a, o = self.encoder.self, self.encoder.output
        ps = [self.pos_embed.LayerNorm.weight, self.pos_embed.LayerNorm.bias, a.query.weight, a.query.bias, a.key.weight, a.key.bias,
              a.value.weight, a.value.bias, o.dense.weight, o.dense.bias, o.LayerNorm.weight, o.LayerNorm.bias]
        if self.out_linear is not None:
            return ps + [self.out_linear.weight, self.out_linear.bias, None, self.pos_embed.position_embeddings.weight]
        return ps + [None, None, self.mod_linear.weight, self.pos_embed.position_embeddings.weight]

    def get(self):
        ps = self._params()
        key = (_PARAM_EPOCH,) + tuple((t.data_ptr(), t._version) for t in ps if t is not None)
        if key != self.key:
            L = native.lib()
            self.blob = torch.empty(L.dldkd_tower_blob_bytes(int(self.out_linear is not None)), dtype=torch.uint8, device=ps[0].device)
            args = [None if t is None else native.ptr(_chk(t.detach().contiguous(), "tower_pack")) for t in ps]
            native.check(L.dldkd_tower_pack_h16(*args, int(ps[-1].shape[0]), native.ptr(self.blob), native.stream()), "tower_pack")
            self.key = key
        return self


def tower_seq_ok(h0):
    """The fused tower serves inference in throughput mode (bf16 GEMMs), hidden 384, at most 128 rows per sequence."""
    return (TOWER_SEQ and _PRECISION == "bf16" and not torch.is_grad_enabled() and h0.is_cuda and h0.shape[-1] == HIDDEN
            and h0.dim() == 3 and h0.shape[1] <= 128)


def plan_tower_items(lens):
    """Host-side packing of 32-row tiles into workgroups of four slots: int32 (n_items, 4) of (seq << 10) | (tile << 8) | length,
    -1 = idle.  A sequence's tiles sit in consecutive slots of one workgroup; sequences of length 0 get no slot."""
    import numpy as np
    lens = np.asarray(lens, dtype=np.int64)
    nt = (lens + 31) // 32
    if (nt > 4).any() or len(lens) >= (1 << 21):
        raise native.NativeError("plan_tower_items: at most 128 rows per sequence and 2^21 sequences")
    by = {k: np.nonzero(nt == k)[0] for k in (1, 2, 3, 4)}
    n1, n2, n3 = len(by[1]), len(by[2]), len(by[3])
    rows = []

    def block(seqs_tiles):          # list of (seq array, n tiles) columns side by side -> (n, 4) int32
        n = len(seqs_tiles[0][0])
        out = np.full((n, 4), -1, np.int64)
        c = 0
        for seqs, k in seqs_tiles:
            for t in range(k):
                out[:, c] = (seqs << 10) | (t << 8) | lens[seqs]
                c += 1
        return out
    if len(by[4]):
        rows.append(block([(by[4], 4)]))
    ones = by[1]
    k31 = min(n3, n1)                                   # 3 + 1
    if k31:
        rows.append(block([(by[3][:k31], 3), (ones[:k31], 1)]))
    if n3 > k31:
        rows.append(block([(by[3][k31:], 3)]))
    ones = ones[k31:]
    k22 = n2 // 2                                       # 2 + 2
    if k22:
        rows.append(block([(by[2][:k22], 2), (by[2][k22:2 * k22], 2)]))
    if n2 % 2:                                          # 2 + 1 + 1
        it = np.full((1, 4), -1, np.int64)
        s = by[2][-1]
        it[0, 0], it[0, 1] = (s << 10) | lens[s], (s << 10) | (1 << 8) | lens[s]
        for c in (2, 3):
            if len(ones):
                it[0, c] = (ones[0] << 10) | lens[ones[0]]
                ones = ones[1:]
        rows.append(it)
    if len(ones):                                       # 1 + 1 + 1 + 1
        pad = (-len(ones)) % 4
        o = np.concatenate([(ones << 10) | lens[ones], np.full(pad, -1, np.int64)]).reshape(-1, 4)
        rows.append(o)
    if not rows:
        return np.zeros((0, 4), np.int32)
    return np.ascontiguousarray(np.concatenate(rows, 0).astype(np.int32))


_NONFINITE = {}


def nonfinite_flag(device):
    """The device word the fused towers (K5) raise when an fp16 operand upstream overflowed (include/dldkd_hip.h,
    dldkd_tower_seq_h16: nonfinite_flag): one per device, zeroed when created and by take_nonfinite()."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    t = _NONFINITE.get(key)
    if t is None:
        t = _NONFINITE[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


def take_nonfinite(device):
    """True when a fused-tower launch since the last call saw a non-finite LayerNorm sum on a valid row (host read: one sync);
    the flag is cleared."""
    t = nonfinite_flag(device)
    hit = bool(int(t.item()))
    if hit:
        t.zero_()
    return hit


def tower_seq(h0, packs, lens, seq_rows=0, row0=None, items=None, out_mode=0, gallery=None, v0=0, Lp=0, lens_out=None,
              skip_zero_rows=False):
    """h0: list (one per branch) of fp32 rows (..., 384) - the input projection's output; packs: list of TowerPack; lens int32
    GPU (n_seq).  items: int32 GPU (n_items, 4) from plan_tower_items or None (workgroup i = sequence i, rows i * seq_rows ..).
    out_mode 0 -> list of fp32 tensors shaped like h0; out_mode 1 -> writes videos v0 .. of the bf16 gallery blobs;
    out_mode 2 (query-tower packs, at most 32 rows per sequence) -> list of (n_seq, 384) modular query vectors."""
    L = native.lib()
    nb = len(h0)
    fs = [p.get() for p in packs]
    if any((f.out_linear is None) != (out_mode == 2) for f in fs):
        raise native.NativeError("tower_seq: out_mode 2 takes query-tower packs, out_mode 0 / 1 video-tower packs")
    h16 = h0[0].dtype == H16
    if h16:
        if out_mode != 1 or row0 is None or any(x.dtype != H16 or not x.is_cuda or not x.is_contiguous() or x.shape[-1] != HIDDEN
                                                for x in h0):
            raise native.NativeError("tower_seq: fp16 h0 rows serve the gallery mode (out_mode 1) with a row0 table only")
        hs = [x.reshape(-1, HIDDEN) for x in h0]
    else:
        hs = [_chk(x.reshape(-1, HIDDEN), "tower_seq.h0") for x in h0]
    if lens.dtype != torch.int32 or not lens.is_cuda:
        raise native.NativeError("tower_seq: lens must be an int32 GPU tensor")
    n_seq = lens.shape[0]
    n_items = items.shape[0] if items is not None else ((n_seq + 3) // 4 if out_mode == 2 else n_seq)
    outs = None
    if out_mode == 0:
        outs = [torch.empty_like(x) for x in hs]
    elif out_mode == 2:
        outs = [torch.empty(n_seq, HIDDEN, dtype=torch.float32, device=lens.device) for _ in hs]
    if h16:
        native.check(L.dldkd_tower_seq_h16_rows16(native.ptr_array(hs), native.ptr_array([f.blob for f in fs]), native.ptr(row0),
                                                native.ptr(lens), native.ptr(items), n_items, n_seq, nb, native.ptr_array(gallery),
                                                int(v0), int(Lp), native.ptr(lens_out), native.ptr(nonfinite_flag(lens.device)),
                                                int(bool(skip_zero_rows)), native.stream()), "tower_seq_h16")
        return None
    native.check(L.dldkd_tower_seq_h16(native.ptr_array(hs), native.ptr_array([f.blob for f in fs]),
                                        native.ptr(row0), native.ptr(lens), native.ptr(items), n_items, n_seq, nb,
                                        out_mode, native.ptr_array(outs) if outs is not None else None, int(seq_rows),
                                        native.ptr_array(gallery) if gallery is not None else None, int(v0), int(Lp),
                                        native.ptr(lens_out), native.ptr(nonfinite_flag(lens.device)), native.stream()), "tower_seq")
    if out_mode == 0:
        return [o.view(x.shape) for o, x in zip(outs, h0)]
    return outs
