"""Deterministic synthetic weights and inputs shared by the golden-vector generator and the tests.

Everything is drawn from numpy's legacy MT19937 `RandomState(seed)` (bit-stable across numpy
versions and machines), so the golden files only need to store OUTPUTS: the generator script
(make_golden.py, build container, reference imported) and the tests (any machine) rebuild
identical inputs from the seed.  Shapes follow SURVEY.md section 8(d).
"""
import numpy as np
import torch

HIDDEN = 384
TEACHER_DIM = 512


def param_shapes(dv, dq, hidden=HIDDEN, max_ctx_l=128, max_desc_l=30, double_branch=True):
    """Name -> shape of the reference state dict (model.py:20-61), in module order."""
    shapes = {}

    def tower(prefix, kind, din, max_l):
        shapes[f"{prefix}{kind}_pos_embed.position_embeddings.weight"] = (max_l, hidden)
        shapes[f"{prefix}{kind}_pos_embed.LayerNorm.weight"] = (hidden,)
        shapes[f"{prefix}{kind}_pos_embed.LayerNorm.bias"] = (hidden,)
        shapes[f"{prefix}{kind}_input_proj.LayerNorm.weight"] = (din,)
        shapes[f"{prefix}{kind}_input_proj.LayerNorm.bias"] = (din,)
        shapes[f"{prefix}{kind}_input_proj.net.1.weight"] = (hidden, din)
        shapes[f"{prefix}{kind}_input_proj.net.1.bias"] = (hidden,)
        for lin in ("self.query", "self.key", "self.value", "output.dense"):
            shapes[f"{prefix}{kind}_encoder.{lin}.weight"] = (hidden, hidden)
            shapes[f"{prefix}{kind}_encoder.{lin}.bias"] = (hidden,)
        shapes[f"{prefix}{kind}_encoder.output.LayerNorm.weight"] = (hidden,)
        shapes[f"{prefix}{kind}_encoder.output.LayerNorm.bias"] = (hidden,)

    for pre in ("", "exp_") if double_branch else ("",):
        tower(pre, "query", dq, max_desc_l)
        shapes[f"{pre}modular_vector_mapping.weight"] = (1, hidden)
        tower(pre, "visual", dv, max_ctx_l)
        shapes[f"{pre}out_mapping_linear.weight"] = (hidden, hidden)
        shapes[f"{pre}out_mapping_linear.bias"] = (hidden,)
    return shapes


def make_params(seed, dv, dq, dtype=torch.float32, **kw):
    """Non-trivial weights: matrices ~N(0, 0.02) like the reference init (model.py:80-93) but
    with non-zero biases and non-unit LayerNorm gains so bias/gain bugs cannot hide."""
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in param_shapes(dv, dq, **kw).items():
        if name.endswith("LayerNorm.weight"):
            a = 1.0 + 0.1 * rs.standard_normal(shape)
        elif name.endswith("LayerNorm.bias"):
            a = 0.05 * rs.standard_normal(shape)
        elif name.endswith(".bias"):
            a = 0.02 * rs.standard_normal(shape)
        else:
            a = 0.02 * rs.standard_normal(shape)
        out[name] = torch.from_numpy(a.astype(np.float64)).to(dtype)
    return out


def _l2norm(a, eps=1e-5):
    # data_provider.py:71-73 adds eps to the norm
    return a / (np.linalg.norm(a, axis=-1, keepdims=True) + eps)


def make_videos(rs, nv, L, dv, lens):
    x = _l2norm(rs.standard_normal((nv, L, dv)))
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.float64)
    return x * mask[:, :, None], mask


def make_texts(rs, nq, lq_max, dq, qlens):
    x = _l2norm(rs.standard_normal((nq, lq_max, dq)))
    mask = (np.arange(lq_max)[None, :] < qlens[:, None]).astype(np.float64)
    return x * mask[:, :, None], mask


def make_train_batch(seed, nv=64, caps=1, L=16, len_lo=4, dv=3072, dq=768, lq_lo=5, lq_hi=30,
                     dtype=torch.float32):
    """A collate_train-shaped batch (data_provider.py:129-136); `caps` captions per video, or a
    list of per-video caption counts (sorted descending like collate_train :116-117)."""
    rs = np.random.RandomState(seed)
    lens = rs.randint(len_lo, L + 1, size=nv)
    lens[0] = L
    counts = [caps] * nv if isinstance(caps, int) else list(caps)
    labels = [i for i, c in enumerate(counts) for _ in range(c)]
    nq = len(labels)
    qlens = rs.randint(lq_lo, lq_hi + 1, size=nq)
    qlens[0] = lq_hi
    sv, vmask = make_videos(rs, nv, L, dv, lens)
    st, tmask = make_texts(rs, nq, lq_hi, dq, qlens)
    tv = 0.3 * rs.standard_normal((nv, L, TEACHER_DIM)) * vmask[:, :, None]
    tt = 0.3 * rs.standard_normal((nq, 1, TEACHER_DIM))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    return dict(student_videos=t(sv), student_videos_mask=t(vmask), teacher_videos=t(tv),
                student_text=t(st), student_text_mask=t(tmask), teacher_text=t(tt),
                text_labels=labels)


# golden G4t (the reference's forward + backward in model.train() with dropout 0, padded to L = 64 with 3..64 valid clips):
# (tag, hard negatives, captions per video, videos, seed)
G4T_CASES = (("t_hard", True, "mixed", 48, 51), ("t_rand", False, 2, 40, 52))


def g4t_caps(caps, nv):
    """Captions per video, sorted descending like collate_train (data_provider.py:116-117)."""
    if caps != "mixed":
        return caps
    return sorted([3] * (nv // 8) + [2] * (nv // 2) + [1] * (nv - nv // 8 - nv // 2), reverse=True)


def g4t_batch(tag, dtype=torch.float32):
    _, hard, caps, nv, seed = next(c for c in G4T_CASES if c[0] == tag)
    return make_train_batch(seed, nv=nv, caps=g4t_caps(caps, nv), L=64, len_lo=3, dv=3072, dq=768, dtype=dtype), hard, nv, seed


def make_eval_sets(seed, nv=64, caps=3, len_lo=4, len_hi=16, dv=3072, dq=768, lq_lo=5, lq_hi=30):
    """In-memory stand-ins for VisDataSet4DLDKD / TxtDataSet4DLDKD (data_provider.py:307-309,
    :344-354): lists of (feat (len,D) float32, index, id)."""
    rs = np.random.RandomState(seed)
    vids, txts = [], []
    for i in range(nv):
        n = int(rs.randint(len_lo, len_hi + 1))
        vids.append((torch.from_numpy(_l2norm(rs.standard_normal((n, dv))).astype(np.float32)), i, f"vid{i:05d}"))
    k = 0
    for i in range(nv):
        for c in range(caps):
            n = int(rs.randint(lq_lo, lq_hi + 1))
            txts.append((torch.from_numpy(_l2norm(rs.standard_normal((n, dq))).astype(np.float32)), k,
                         f"vid{i:05d}#enc#{c}"))
            k += 1
    return vids, txts


class ListDataset(torch.utils.data.Dataset):
    def __init__(self, items):
        self.items = items

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def make_gallery(seed, nq, nv, L, len_lo, d=HIDDEN, sigma=0.0, dtype=torch.float32, device="cpu"):
    """Directly synthesised ENCODED gallery + queries for scoring-only runs:
    g ~ randn(Nv,L,D), lens ~ U{len_lo..L}; query m is planted on a random valid clip of video
    (m mod Nv): q = g[gt, l*] + sigma * ||g|| / sqrt(D) * randn (sigma=0 -> pure randn queries
    when plant=False).  Returns dict(q, g, lens, mask, gt)."""
    gen = torch.Generator(device="cpu").manual_seed(seed)
    g = torch.randn(nv, L, d, generator=gen, dtype=torch.float32)
    lens = torch.randint(len_lo, L + 1, (nv,), generator=gen)
    lens[0] = L
    mask = (torch.arange(L).unsqueeze(0) < lens.unsqueeze(1)).float()
    g = g * mask.unsqueeze(-1)
    gt = torch.arange(nq) % nv
    lstar = (torch.rand(nq, generator=gen) * lens[gt].float()).long().clamp(max=L - 1)
    base = g[gt, lstar]
    q = base + sigma * torch.randn(nq, d, generator=gen)
    return dict(q=q.to(dtype).to(device), g=g.to(dtype).to(device), lens=lens.to(torch.int32).to(device),
                mask=mask.to(dtype).to(device), gt=gt.to(device))
