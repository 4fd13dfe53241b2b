"""CPU oracle for the DL-DKD scoring + distillation hot path.   *** TEST INFRASTRUCTURE ONLY ***

This module restates, on the CPU and in plain torch/numpy, what the upstream reference
(HuiGuanLab/DL-DKD, files method/model.py, method/model_components.py, method/eval.py,
method/optimization.py) computes on the path named by BASELINE.json's north_star.  It is the
checker for the HIP kernels; it is NOT a product path.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.  The shipped package (dl-dkd_amd/) never
imports it and raises when its HIP library is missing.

Parity pinning: every function here is checked against the reference itself (imported in the
build container) by tests/golden/make_golden.py, and against the committed outputs of that run
(tests/golden/*.npz) by tests/test_oracle_golden.py.  The reference ships no tests or golden
vectors of its own, so the pins are "outputs of the reference itself run
here", as the task allows.

All arithmetic is floating point.  Functions take a `p` dict of tensors keyed by the
reference's state-dict names (74 keys, model.py:20-61) and work in whatever dtype `p` holds
(fp32 for parity with the reference, fp64 when used as a tighter yardstick for kernels).
The restatement is vectorised where the reference loops in Python; sums are therefore
re-associated and agree with the reference to fp32 rounding, not bitwise.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-5            # nn.LayerNorm default, model_components.py:274,301,443
MASK_FILL = -1e10        # model.py:444-445
ATTN_MASK_FILL = -10000.0  # model_components.py:422
KL_TEMP = 0.2            # model.py:155


# --------------------------------------------------------------------------------------
# encoder towers
# --------------------------------------------------------------------------------------
def _ln(x, w, b):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + LN_EPS) * w + b


def input_projection(x, p, prefix):
    """LayerNorm(Din) -> Linear(Din,D) -> ReLU (dropout is identity in eval).

    Follows LinearLayer.forward, model_components.py:305-312.
    """
    h = _ln(x, p[prefix + ".LayerNorm.weight"], p[prefix + ".LayerNorm.bias"])
    h = h @ p[prefix + ".net.1.weight"].t() + p[prefix + ".net.1.bias"]
    return torch.relu(h)


def add_position(x, p, prefix):
    """x + learned position rows [0, L) then LayerNorm.

    Follows TrainablePositionalEncoding.forward, model_components.py:277-284.
    """
    L = x.shape[1]
    pos = p[prefix + ".position_embeddings.weight"][:L]
    return _ln(x + pos.unsqueeze(0), p[prefix + ".LayerNorm.weight"], p[prefix + ".LayerNorm.bias"])


def self_attention(x, mask, p, prefix, n_heads):
    """Multi-head self attention with an additive -10000 key mask.

    Follows BertSelfAttention.forward, model_components.py:398-436 (scale 1/sqrt(dh) :419,
    mask :422, softmax over keys :426, context :432-435).
    """
    N, L, D = x.shape
    dh = D // n_heads

    def split(t):
        return t.view(N, L, n_heads, dh).permute(0, 2, 1, 3)

    q = split(x @ p[prefix + ".query.weight"].t() + p[prefix + ".query.bias"])
    k = split(x @ p[prefix + ".key.weight"].t() + p[prefix + ".key.bias"])
    v = split(x @ p[prefix + ".value.weight"].t() + p[prefix + ".value.bias"])
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    if mask is not None:
        s = s + ((1 - mask) * ATTN_MASK_FILL).view(N, 1, 1, L)
    a = torch.softmax(s, dim=-1)
    ctx = (a @ v).permute(0, 2, 1, 3).reshape(N, L, D)
    return ctx


def attention_block(x, mask, p, prefix, n_heads):
    """BertAttention = self attention + (dense, residual, LayerNorm).

    Follows BertAttention.forward model_components.py:345-353 and BertSelfOutput.forward
    :446-450.
    """
    ctx = self_attention(x, mask, p, prefix + ".self", n_heads)
    h = ctx @ p[prefix + ".output.dense.weight"].t() + p[prefix + ".output.dense.bias"]
    return _ln(h + x, p[prefix + ".output.LayerNorm.weight"], p[prefix + ".output.LayerNorm.bias"])


def encode_input(feat, mask, p, proj, enc, pos, n_heads):
    """Tower composition, model.py:229-243."""
    h = input_projection(feat, p, proj)
    h = add_position(h, p, pos)
    return attention_block(h, mask, p, enc, n_heads)


def encode_context(p, feat, mask, n_heads=4, double_branch=True):
    """Video towers of both branches, model.py:215-227 -> (inh (Nv,L,D), exp or None)."""
    out = []
    for pre in ("", "exp_") if double_branch else ("",):
        h = encode_input(feat, mask, p, pre + "visual_input_proj", pre + "visual_encoder",
                         pre + "visual_pos_embed", n_heads)
        h = h @ p[pre + "out_mapping_linear.weight"].t() + p[pre + "out_mapping_linear.bias"]
        out.append(h)
    return (out[0], out[1]) if double_branch else (out[0], None)


def modular_pool(h, mask, w):
    """Attention pooling of the word states into one query vector.

    Follows get_modularized_queries, model.py:245-258: logits = h.w, masked words get exactly
    -1e10 (mask_logits :444-445), softmax over words, weighted sum.  The reference's trailing
    .squeeze() (which would also drop a batch dim of 1) is restated as dropping only the
    singleton "module" dim.
    """
    logits = (h @ w.t()).squeeze(-1)                      # (N, L)
    logits = logits * mask + (1 - mask) * MASK_FILL
    a = torch.softmax(logits, dim=1)
    return torch.einsum("nl,nld->nd", a, h)


def encode_query(p, feat, mask, n_heads=4, double_branch=True):
    """Query towers, model.py:199-211 -> (inh (Nq,D), exp or None)."""
    out = []
    for pre in ("", "exp_") if double_branch else ("",):
        h = encode_input(feat, mask, p, pre + "query_input_proj", pre + "query_encoder",
                         pre + "query_pos_embed", n_heads)
        out.append(modular_pool(h, mask, p[pre + "modular_vector_mapping.weight"]))
    return (out[0], out[1]) if double_branch else (out[0], None)


# --------------------------------------------------------------------------------------
# similarity + key-clip max-pool
# --------------------------------------------------------------------------------------
def _clip_scores(q, ctx, mask):
    s = torch.einsum("md,nld->mln", q, ctx)               # (Nq, L, Nv): video index LAST
    if mask is not None:
        m = mask.transpose(0, 1).unsqueeze(0)             # (1, L, Nv)
        s = s * m + (1 - m) * MASK_FILL
    return s


def sim_scores(q, ctx, mask=None):
    """Cosine clip scores and their max over clips.

    Follows get_sim_scores, model.py:307-329: F.normalize (eps 1e-12) both sides :318-319,
    einsum "md,nld->mln" :321, mask_logits :325, max over dim 1 :327.
    Returns (pooled (Nq,Nv), clip_level (Nq,L,Nv), argmax (Nq,Nv)).
    """
    qn = F.normalize(q, dim=-1)
    cn = F.normalize(ctx, dim=-1)
    s = _clip_scores(qn, cn, mask)
    pooled, idx = s.max(dim=1)
    return pooled, s, idx


def unnormalized_sim_scores(q, ctx, mask=None):
    """Raw dot-product twin of sim_scores, model.py:331-350.  Returns pooled (Nq,Nv)."""
    s = _clip_scores(q, ctx, mask)
    return s.max(dim=1)[0]


def eval_scores(q_inh, q_exp, g_inh, g_exp, mask, chunk=50):
    """Score matrix as compute_query2ctx_info builds it (eval.py:188-212): queries in chunks
    of eval_query_bsz, both branches, pooled cosine scores only.  Returns (inh, exp) (Nq,Nv)."""
    inh, exp = [], []
    for s in range(0, q_inh.shape[0], chunk):
        inh.append(sim_scores(q_inh[s:s + chunk], g_inh, mask)[0])
        if q_exp is not None:
            exp.append(sim_scores(q_exp[s:s + chunk], g_exp, mask)[0])
    return torch.cat(inh, 0), (torch.cat(exp, 0) if exp else None)


def fuse_scores(inh, exp):
    """0.7 / 0.3 branch fusion, eval.py:254."""
    return 0.7 * inh + 0.3 * exp


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------
def kl_frame_score(predict, target, mask, labels, temp=KL_TEMP):
    """Teacher-vs-student KL over the clips of each query's own video, summed over queries.

    Follows compute_kl_loss(mode='frame_score'), model.py:183-197: column predict[i,:,label_i],
    first len_v clips (len = #mask>0 of that video), KL(softmax(t/temp) || softmax(p/temp)),
    reduction 'sum', accumulated over i.
    predict/target: (Nq, L, Nv) clip scores; mask (Nv, L); labels: Nq ints.
    """
    lab = torch.as_tensor(labels, dtype=torch.long)
    idx = torch.arange(len(lab))
    p = predict[idx, :, lab]                               # (Nq, L)
    t = target[idx, :, lab]
    lens = (mask[lab] > 0).sum(1)                          # (Nq,)
    valid = torch.arange(p.shape[1]).unsqueeze(0) < lens.unsqueeze(1)
    neg_inf = torch.finfo(p.dtype).min
    logp = torch.log_softmax(torch.where(valid, p / temp, torch.full_like(p, neg_inf)), dim=-1)
    tt = torch.where(valid, t / temp, torch.full_like(t, neg_inf))
    logt = torch.log_softmax(tt, dim=-1)
    tp = torch.exp(logt)
    term = torch.where(valid & (tp > 0), tp * (logt - logp), torch.zeros_like(tp))
    return term.sum()


def draw_triplet_randoms(labels, n_videos, use_hard_negative, hard_pool_size):
    """Consume torch's global CPU RNG exactly as one get_clip_triplet_loss call does
    (model.py:366-368 then :377-380) and return the draws.

    non-hard: one torch.randint(0, n_neg_i, (1,)) per video i, then one
    torch.randint(1, Nv, (Nq,)); hard: only torch.randint(1, min(1+pool, Nv), (Nq,)).
    """
    lab = np.asarray(labels)
    r_v2t = None
    if not use_hard_negative:
        r_v2t = []
        for i in range(n_videos):
            n_neg = int((lab != i).sum())
            r_v2t.append(int(torch.randint(0, n_neg, size=(1,))))
        r_v2t = torch.tensor(r_v2t, dtype=torch.long)
    hi = min(1 + hard_pool_size, n_videos) if use_hard_negative else n_videos
    r_t2v = torch.randint(1, hi, size=(len(lab),))
    return r_v2t, r_t2v


def clip_triplet_loss(scores, labels, margin, use_hard_negative, r_v2t, r_t2v):
    """Video->text + text->video hinge loss on pooled cosine scores.

    Follows get_clip_triplet_loss, model.py:353-387.  v2t (:360-369): per video, mean of its
    own queries' scores vs the hardest (or the r_v2t[i]-th, descending) other-query score.
    t2v (:372-385): per query, positive vs the r_t2v[q]-th entry of its row sorted descending
    with the positive forced to rank 0.  `r_*` are the integer draws (see
    draw_triplet_randoms); r_v2t is ignored in hard mode.
    """
    lab = torch.as_tensor(labels, dtype=torch.long)
    nq, nv = scores.shape
    own = lab.unsqueeze(1) == torch.arange(nv).unsqueeze(0)            # (Nq, Nv)
    zero = torch.zeros((), dtype=scores.dtype)
    # v2t
    cnt = own.sum(0).to(scores.dtype)
    pos_v = torch.where(own, scores, zero).sum(0) / cnt                # nan if a video has no query
    neg_inf = torch.finfo(scores.dtype).min
    others = torch.where(own, torch.full_like(scores, neg_inf), scores)  # (Nq, Nv)
    sorted_others, _ = torch.sort(others.t(), dim=1, descending=True)  # per video
    if use_hard_negative:
        neg_v = sorted_others[:, 0]
    else:
        neg_v = sorted_others[torch.arange(nv), r_v2t]
    v2t = torch.clamp(margin + neg_v - pos_v, min=0).sum()
    # t2v
    qi = torch.arange(nq)
    pos_q = scores[qi, lab]
    masked = scores.detach().clone()
    masked[qi, lab] = 999
    order = torch.sort(masked, descending=True, dim=1)[1]
    neg_q = scores[qi, order[qi, r_t2v]]
    t2v = torch.clamp(margin + neg_q - pos_q, min=0).sum()
    return t2v / nq + v2t / nv


def nce_soft(labels, scores, sims, alpha, beta):
    """Soft-label symmetric InfoNCE.

    Follows clip_nce_soft.forward, model_components.py:126-199 (reduction='mean'):
    rows q >= floor(alpha*Nq) / videos v >= floor(alpha*Nv) get soft targets
    clamp((1-beta)*softmax(sims) + beta*onehot, 0); hard part uses the one-hot rows.
    t2v part  = sum_q sum_v I_Q[q,v] * (LSE_v S[q,:] - S[q,v])
    v2t part  = sum_v [LSE_q S[:,v] - LSE_q(log(I_V[v,q] + 1e-12) + S[q,v])], over videos that
    own at least one query (label_dict.items() loops, :169-180).
    """
    lab = torch.as_tensor(labels, dtype=torch.long)
    nq, nv = scores.shape
    hard_q = math.floor(alpha * nq)
    hard_v = math.floor(alpha * nv)
    soft_q, soft_v = nq - hard_q, nv - hard_v
    onehot = (lab.unsqueeze(1) == torch.arange(nv).unsqueeze(0)).to(scores.dtype)   # (Nq,Nv)
    has_query = onehot.sum(0) > 0

    i_q = onehot.clone()
    sq = torch.softmax(sims, dim=-1)
    i_q[hard_q:] = torch.clamp((1 - beta) * sq[hard_q:] + beta * onehot[hard_q:], min=0)
    i_v = onehot.t().clone()                                # (Nv, Nq)
    sv = torch.softmax(sims.t(), dim=-1)
    i_v[hard_v:] = torch.clamp((1 - beta) * sv[hard_v:] + beta * onehot.t()[hard_v:], min=0)

    lse_row = torch.logsumexp(scores, dim=1, keepdim=True)  # (Nq,1)
    t2v_rows = (i_q * (lse_row - scores)).sum(1)            # (Nq,)
    t2v_hard, t2v_soft = t2v_rows[:hard_q].sum(), t2v_rows[hard_q:].sum()

    lse_col = torch.logsumexp(scores, dim=0)                # (Nv,)
    nom = torch.logsumexp(torch.log(i_v + 1e-12) + scores.t(), dim=1)   # (Nv,)
    v2t_cols = torch.where(has_query, lse_col - nom, torch.zeros_like(nom))
    v2t_hard, v2t_soft = v2t_cols[:hard_v].sum(), v2t_cols[hard_v:].sum()

    hard_loss = 0.0
    soft_loss = 0.0
    if hard_q != 0 and hard_v != 0:
        hard_loss = t2v_hard / hard_q + v2t_hard / hard_v
    if soft_q != 0 and soft_v != 0:
        soft_loss = t2v_soft / soft_q + v2t_soft / soft_v
    return alpha * hard_loss + (1 - alpha) * soft_loss


def nce_hard(labels, scores):
    """Hard-label symmetric InfoNCE, clip_nce.forward model_components.py:216-234:
    mean_q(LSE_v S[q,:] - S[q,label_q]) + mean_v(LSE_q S[:,v] - LSE_{q in pos(v)} S[q,v]);
    videos without a query contribute 0 - 0 (both buffers stay zero, :226-232)."""
    lab = torch.as_tensor(labels, dtype=torch.long)
    nq, nv = scores.shape
    own = lab.unsqueeze(1) == torch.arange(nv).unsqueeze(0)
    t2v = torch.logsumexp(scores, dim=1) - scores[torch.arange(nq), lab]
    neg_inf = torch.finfo(scores.dtype).min
    nom = torch.logsumexp(torch.where(own, scores, torch.full_like(scores, neg_inf)), dim=0)
    den = torch.logsumexp(scores, dim=0)
    has_query = own.any(0)
    v2t = torch.where(has_query, den - nom, torch.zeros_like(den))
    return t2v.mean() + v2t.mean()


def forward_losses(p, batch, cfg, triplet_randoms):
    """The whole training forward, DLDKD.forward model.py:100-163, dropout off.

    cfg: dict with n_heads, margin, use_hard_negative, label_style, kl_intra_weight, weight,
    inher_nce_weight, explore_nce_weight, alpha, belta.
    triplet_randoms: ((r_v2t, r_t2v) for inheritance, (r_v2t, r_t2v) for exploration) in the
    order the reference draws them (:137 then :147).
    Returns dict of the 7 loss entries (tensors) with 'loss' the total.
    """
    labels = batch["text_labels"]
    mask = batch["student_videos_mask"]
    g_inh, g_exp = encode_context(p, batch["student_videos"], mask, cfg["n_heads"])
    q_inh, q_exp = encode_query(p, batch["student_text"], batch["student_text_mask"], cfg["n_heads"])
    t_text = batch["teacher_text"].reshape(batch["teacher_text"].shape[0], -1)   # .squeeze() :114

    _, t_clip, _ = sim_scores(t_text, batch["teacher_videos"], mask)
    t_raw = unnormalized_sim_scores(t_text, batch["teacher_videos"], mask)
    i_cos, i_clip, _ = sim_scores(q_inh, g_inh, mask)
    i_raw = unnormalized_sim_scores(q_inh, g_inh, mask)
    e_cos, _, _ = sim_scores(q_exp, g_exp, mask)
    e_raw = unnormalized_sim_scores(q_exp, g_exp, mask)

    hard = cfg["use_hard_negative"]
    inher_trip = clip_triplet_loss(i_cos, labels, cfg["margin"], hard, *triplet_randoms[0])
    explore_trip = clip_triplet_loss(e_cos, labels, cfg["margin"], hard, *triplet_randoms[1])
    if cfg["label_style"] == "soft":
        inher_nce = cfg["inher_nce_weight"] * nce_soft(labels, i_raw, t_raw, cfg["alpha"], cfg["belta"])
        explore_nce = cfg["explore_nce_weight"] * nce_soft(labels, e_raw, e_raw, cfg["alpha"], cfg["belta"])
    else:
        inher_nce = cfg["inher_nce_weight"] * nce_hard(labels, i_raw)
        explore_nce = cfg["explore_nce_weight"] * nce_hard(labels, e_raw)
    kl_intra = cfg["kl_intra_weight"] * cfg["weight"] * kl_frame_score(i_clip, t_clip, mask, labels)
    loss = inher_trip + inher_nce + kl_intra + explore_trip + explore_nce
    return dict(loss=loss, inher_trip=inher_trip, inher_nce=inher_nce, explore_trip=explore_trip,
                explore_nce=explore_nce, kl=kl_intra, kl_intra=kl_intra)


# --------------------------------------------------------------------------------------
# eval: ground truth, ranking, R@K
# --------------------------------------------------------------------------------------
def get_gt(video_metas, query_metas):
    """cap_id 'vid#...' belongs to video 'vid' (eval.py:43-57).  Returns (v2t_gt, t2v_gt)."""
    pos = {v: i for i, v in enumerate(video_metas)}
    v2t = [[] for _ in video_metas]
    t2v = {}
    for qi, cap in enumerate(query_metas):
        vi = pos.get(cap.split("#", 1)[0])
        if vi is not None:
            v2t[vi].append(qi)
    for vi, qs in enumerate(v2t):                           # same insertion order as :52-55
        for qi in qs:
            t2v.setdefault(qi, []).append(vi)
    return v2t, t2v


def gt_ranks(errors, t2v_gt):
    """Best rank (1-based) of any GT video per query, by ascending sort of the error matrix
    (= -score), eval.py:69-83.  Ties: counted optimistically (rank = 1 + #strictly better);
    the reference's np.argsort breaks them arbitrarily."""
    n_q = errors.shape[0]
    ranks = np.zeros((n_q,), np.int64)
    for i in range(n_q):
        best = min(errors[i, k] for k in t2v_gt[i])
        ranks[i] = 1 + int((errors[i] < best).sum())
    return ranks


def eval_q2m(errors, t2v_gt):
    """(R@1, R@5, R@10, R@100, MedR, MeanR), eval.py:59-94."""
    r = gt_ranks(errors, t2v_gt)
    n_q = errors.shape[0]
    rk = [100.0 * int((r <= k).sum()) / n_q for k in (1, 5, 10, 100)]
    return (rk[0], rk[1], rk[2], rk[3], float(np.median(r)), float(r.mean()))


def t2v_map(errors, t2v_gt):
    """Mean AP using only the FIRST GT video of each query (eval.py:97-111, quirk :106):
    with one relevant item AP = 1 / rank of that item."""
    n_q = errors.shape[0]
    ap = np.zeros(n_q)
    for i in range(n_q):
        x = t2v_gt[i][0]
        rank = 1 + int((errors[i] < errors[i, x]).sum())
        ap[i] = 1.0 / rank
    return float(ap.mean())


def eval_metrics(inh, exp, video_metas, query_metas):
    """eval_epoch's tail (eval.py:246-263): three rankings, SumR of the fused one."""
    _, t2v = get_gt(video_metas, query_metas)
    out = {"inher": eval_q2m(-1 * inh, t2v)}
    if exp is not None:
        out["explore"] = eval_q2m(-1 * exp, t2v)
        fused = 0.7 * inh + 0.3 * exp
    else:
        fused = inh
    out["fused"] = eval_q2m(-1 * fused, t2v)
    out["map"] = t2v_map(-1 * fused, t2v)
    out["sumr"] = sum(out["fused"][:4])
    return out


# --------------------------------------------------------------------------------------
# optimiser (next-row f2)
# --------------------------------------------------------------------------------------
def warmup_linear(step, t_total, warmup):
    """WarmupLinearSchedule.get_lr, optimization.py:121-135,172-175 (multiplier; 0 at step 0)."""
    if t_total < 0:
        return 1.0
    progress = float(step) / t_total
    if progress < warmup:
        return progress / warmup
    return max((progress - 1.0) / (warmup - 1.0), 0.0)


def bert_adam_step(param, grad, m, v, step, lr, weight_decay, t_total, warmup,
                   b1=0.9, b2=0.999, eps=1e-6, max_grad_norm=1.0):
    """One BertAdam update of ONE tensor, optimization.py:296-336: per-tensor clip to
    max_grad_norm (torch clip_grad_norm_: coef = max_norm/(norm+1e-6) clamped to 1), moments
    without bias correction, decoupled weight decay, scheduled lr.  Returns (param, m, v)."""
    g = grad
    if max_grad_norm > 0:
        norm = torch.linalg.vector_norm(g)
        coef = torch.clamp(max_grad_norm / (norm + 1e-6), max=1.0)
        g = g * coef
    m = m * b1 + (1 - b1) * g
    v = v * b2 + (1 - b2) * g * g
    update = m / (v.sqrt() + eps)
    if weight_decay > 0:
        update = update + weight_decay * param
    lr_t = lr * warmup_linear(step, t_total, warmup)
    return param - lr_t * update, m, v


# --------------------------------------------------------------------------------------
# feature ingest (next-row f3)
# --------------------------------------------------------------------------------------
def sampling_bounds(num_clips, max_len):
    """Segment bounds of uniform_feature_sampling (data_provider.py:52-68): idxs = round(arange(max_len+1) /
    max_len * num_clips) with numpy's round-half-to-even, clipped to num_clips-1.  Returns (start, end) int arrays
    of length max_len; segment i is the mean of rows [start, end) when start < end, else the single row start."""
    idxs = np.arange(0, max_len + 1, 1.0) / max_len * num_clips
    idxs = np.round(idxs).astype(np.int32)
    idxs[idxs > num_clips - 1] = num_clips - 1
    return idxs[:-1].copy(), idxs[1:].copy()


def uniform_feature_sampling(features, max_len):
    """Temporal down-sampling to at most max_len clips by segment means (data_provider.py:52-68)."""
    n = features.shape[0]
    if max_len is None or n <= max_len:
        return features
    s, e = sampling_bounds(n, max_len)
    out = np.empty((max_len, features.shape[1]), features.dtype)
    for i in range(max_len):
        out[i] = features[s[i]:e[i]].mean(0) if s[i] < e[i] else features[s[i]]
    return out


def l2_normalize_rows(a, eps=1e-5):
    """x / (||x|| + eps) (data_provider.py:71-73: eps is ADDED to the norm)."""
    return a / (np.linalg.norm(a, axis=-1, keepdims=True) + eps)
