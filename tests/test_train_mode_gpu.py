"""GPU: the TRAIN-mode step - the code path the C3 / C5 step numbers of bench.py are timed on - against the reference and the oracle.

model.train() switches on what model.eval() never runs: the padding-skip input projection (functional._InProjTrain with the batch's
row mask, 32-row group flags), the row-group filters of every linear / LayerNorm / weight-gradient kernel of the video towers, and -
under train.GraphedTrainStep - the multi-graph stepper (a graph per tower and pass on its own stream).  With drop = input_drop = 0
the reference's nn.Dropout is the identity, so its train-mode step is deterministic (hard part 2(a)) and the
oracle (which has no train / eval switch) restates it: golden G4t pins that (tests/test_oracle_golden.py).

  * G4t: reference-generated 7 losses + 74 gradients, batch padded to L = 64 with 3..64 valid clips (whole 32-row groups of padding)
  * C3 / C5 size: 7 losses AND all 74 gradients against the oracle's autograd, parity mode 1e-4 / 1e-3, bf16 mode at its tolerance
  * replayed steps of the multi-graph stepper + fused BertAdam against oracle losses + oracle BertAdam updates (G6-style)
"""
import types

import numpy as np
import pytest
import torch

import dldkd_oracle as orc
import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LOSS_KEYS = ("inher_trip", "inher_nce", "explore_trip", "explore_nce", "kl_intra")


def _train_model(dv, dq, params, hard, drop=0.0):
    from dldkd_amd.model import DLDKD
    cfg = types.SimpleNamespace(visual_input_size=dv, query_input_size=dq, inheritance_hidden=384, exploration_hidden=384,
                                max_ctx_l=128, max_desc_l=30, input_drop=drop, drop=drop, n_heads=4, initializer_range=0.02,
                                margin=0.1, use_hard_negative=hard, hard_pool_size=20, label_style="soft")
    opt = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                collection="tvr", alpha=0.8, belta=0.8)
    m = DLDKD(cfg, opt)
    m.load_state_dict(params, strict=True)
    m = m.to(DEV).train()
    assert m.training and m.visual_input_proj.training
    return m


def _cfg(hard, weight):
    return dict(n_heads=4, margin=0.1, use_hard_negative=hard, label_style="soft", kl_intra_weight=0.1, weight=weight,
                inher_nce_weight=0.04, explore_nce_weight=0.04, alpha=0.8, belta=0.8)


class _FlagLog:
    """Records the row-group flags every training tower publishes (ops.set_row_groups) during a forward pass."""

    def __init__(self, monkeypatch):
        from dldkd_amd import ops
        self.calls = []
        real = ops.set_row_groups

        def spy(flags, M):
            if flags is not None:
                self.calls.append((flags, int(M)))
            return real(flags, M)
        monkeypatch.setattr(ops, "set_row_groups", spy)
        # throughput mode: the fused training towers (functional._TowerTrain) take the flags as an argument instead
        from dldkd_amd import functional as F_
        real_tt = F_.tower_train

        def spy_tt(y0, *a, **k):
            flags = a[10] if len(a) > 10 else k.get("flags")
            if flags is not None:
                self.calls.append((flags, int(y0.shape[0] * y0.shape[1])))
            return real_tt(y0, *a, **k)
        monkeypatch.setattr(F_, "tower_train", spy_tt)

    def assert_skipped(self, lens, L):
        """Both video towers published flags, and they are exactly `group holds a valid clip`."""
        assert len(self.calls) >= 2, "the training video towers did not take the padding-skip path"
        want = (np.arange(0, L, 32)[None, :] < np.asarray(lens)[:, None]).reshape(-1)
        assert not want.all(), "the batch has no all-padding 32-row group: nothing would be skipped"
        n = 0
        for flags, M in self.calls:
            if M == len(lens) * L:
                assert (flags.cpu().numpy().astype(bool) == want).all()
                n += 1
        assert n >= 2
        return int((~want).sum()), want.size


def _grad_stats(model, ref_grads):
    """Per parameter: l2 error relative to max(||ref||, 1e-4 max_t ||ref_t||) and the cosine.  Key biases have an exactly-zero
    gradient (softmax shift invariance): the floor keeps their rounding noise from reading as error."""
    nmax = max(float(r.norm()) for r in ref_grads.values())
    out = {}
    for n, prm in model.named_parameters():
        assert prm.grad is not None, n
        g = prm.grad.detach().double().cpu().reshape(-1)
        r = ref_grads[n].double().reshape(-1)
        assert torch.isfinite(g).all(), n
        rel = float((g - r).norm()) / max(float(r.norm()), 1e-4 * nmax)
        cos = float(g @ r) / max(float(g.norm()) * float(r.norm()), 1e-300)
        out[n] = (rel, cos, float(r.norm()) / nmax)
    return out


# worst tensor's relative l2 gradient error against the oracle's autograd.  Throughput (bf16) mode: measured 0.127 (C3) / 0.164 (C5,
# exp_query_encoder.self.value.bias), median 0.06 - 0.08; unchanged to three digits by round 6's move of the ReLU mask out of bit 0 of
# the saved normalised rows (that bit was not what limits it - every operand of every product is rounded to 8 bits).
# Mixed mode (fp32-grade forward, bf16 backward on exact activations): measured 8.4e-3 / 7.4e-3.
GRAD_TOL_BF16 = 0.20
GRAD_TOL_MIXED = 0.02


def _assert_grads(stats, rel_tol, what, cos_min=None):
    worst = max(stats.items(), key=lambda kv: kv[1][0])
    print(f"  {what}: worst gradient rel l2 error {worst[1][0]:.3e} ({worst[0]}), median "
          f"{float(np.median([v[0] for v in stats.values()])):.3e}, min cosine over tensors with norm > 1e-3 of the largest "
          f"{min([v[1] for v in stats.values() if v[2] > 1e-3]):.5f}")
    bad = {n: v for n, v in stats.items() if v[0] > rel_tol}
    assert not bad, (what, bad)
    if cos_min is not None:
        low = {n: v for n, v in stats.items() if v[2] > 1e-3 and v[1] < cos_min}
        assert not low, (what, low)


# ----------------------------------------------------------------------------------------------------------------- golden G4t
@pytest.mark.parametrize("prec", ["fp32", "bf16", "mixed"])
@pytest.mark.parametrize("tag", [c[0] for c in synth.G4T_CASES])
def test_train_mode_forward_backward_vs_golden_g4t(golden_dir, monkeypatch, tag, prec):
    """The REFERENCE's own train-mode step (model.train(), dropout 0): 7 losses within 1e-4 and all 74 gradients in parity mode,
    2e-2 in throughput (bf16) mode, 1e-4 on the losses with bf16-backward gradients in mixed mode; the padding-skip path must
    actually have skipped the all-padding row groups."""
    from dldkd_amd import ops
    g = np.load(f"{golden_dir}/g4t_forward_train.npz")
    batch, hard, nv, seed = synth.g4t_batch(tag)
    m = _train_model(3072, 768, synth.make_params(seed, 3072, 768), hard)
    m.weight = 0.95 ** 3
    lens = g[f"{tag}_lens"]
    dbatch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    log = _FlagLog(monkeypatch)
    ops.set_gemm_precision(prec)
    try:
        torch.manual_seed(777)                            # the CPU RNG state the reference had (triplet negatives)
        loss, d = m(dbatch)
        m.zero_grad()
        loss.backward()
    finally:
        ops.set_gemm_precision("fp32")
    skipped, groups = log.assert_skipped(lens, 64)
    print(f"  {tag} {prec}: {skipped} of {groups} 32-row groups skipped")
    exact_fwd = prec in ("fp32", "mixed")
    tol = 1e-4 if exact_fwd else 2e-2
    for k in LOSS_KEYS + ("loss",):
        got = float(loss if k == "loss" else d[k])
        ref = float(g[f"{tag}_{k}"].reshape(()))
        assert abs(got - ref) <= tol * max(abs(ref), 1e-3 if exact_fwd else 0.1), (k, got, ref)
    names = [n for n, _ in m.named_parameters()]
    assert len(names) == 74
    gmax = max(float(np.abs(g[f"{tag}_grad/{n}/sample"]).max()) for n in names)
    nmax = max(float(g[f"{tag}_grad/{n}/norm"]) for n in names)
    # throughput mode: bf16 operand rounding through the stacked GEMMs; single sampled elements of a small tensor move by tens of
    # per cent of the tensor's largest element while its norm stays within a few per cent (measured 0.26 / 0.37 and 0.03)
    gtol, ntol = (3e-3, 3e-3) if prec == "fp32" else (0.1, 0.03) if prec == "mixed" else (0.6, 0.15)
    worst, worst_n = 0.0, 0.0
    for n, prm in m.named_parameters():
        gr = prm.grad.detach().reshape(-1).cpu()
        idx = np.unique(np.linspace(0, gr.numel() - 1, min(48, gr.numel())).astype(np.int64))
        ref = g[f"{tag}_grad/{n}/sample"].astype(np.float64)
        scale = max(np.abs(ref).max(), 1e-4 * gmax)
        e = np.abs(gr[idx].double().numpy() - ref).max() / scale
        rn = float(g[f"{tag}_grad/{n}/norm"])
        en = abs(float(gr.double().norm()) - rn) / max(rn, 1e-4 * nmax)
        worst, worst_n = max(worst, e), max(worst_n, en)
        assert e <= gtol and en <= ntol, (n, e, en)
    print(f"  {tag} {prec}: worst sampled-gradient error {worst:.3e}, worst norm error {worst_n:.3e}")


# ------------------------------------------------------------------------------------------------- C3 / C5 size, oracle autograd
_ORACLE_CACHE = {}


def _size_case(name):
    """(params, batch, hard, dv, dq, nv): BASELINE configs[2] (TVR step) / configs[4] (one rank's Charades step)."""
    if name == "c3":
        dv, dq, nv = 3072, 768, 128
        return synth.make_params(43, dv, dq), synth.make_train_batch(3, nv=nv, caps=5, L=128, len_lo=24, dv=dv, dq=dq), True, dv, dq, nv
    dv, dq, nv = 1024, 1024, 128
    caps = sorted([3] + [2] * 127, reverse=True)
    return synth.make_params(45, dv, dq), synth.make_train_batch(5, nv=nv, caps=caps, L=64, len_lo=8, dv=dv, dq=dq), True, dv, dq, nv


NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")
N_STEPS, LR, WARMUP, STEPS_PER_EPOCH, N_EPOCH, SEED0 = 3, 3e-4, 0.01, 10, 5, 1000


def _oracle_trajectory(name):
    """N_STEPS of (oracle forward + autograd, oracle BertAdam with the weight-decay groups of train.py:203-213 and its warm-up
    schedule) on ONE batch, CPU draws of torch.manual_seed(SEED0 + step) per step.  Returns (per-step loss dicts, the gradients
    of step 0, the parameters after the last step).  Cached: the parity and bf16 runs check against the same numbers.

    The oracle runs in fp32, the reference's arithmetic (pinned at <= 2e-5 by goldens G4 / G4t).  Measured at the C3 batch: an fp64
    oracle takes one discrete decision of the step (a hard-negative pick / arg-max clip / hinge within fp32 rounding of its
    boundary) the other way than fp32 arithmetic does - the fp32 oracle and the HIP path both sit 3.503e-3 (relative l2, same
    digits) from the fp64 gradient of visual_encoder.self.value.weight and 2.3e-4 worst / 2.8e-6 median from each other - so fp64 is the wrong yardstick
    for 1e-4-grade gradient parity with an fp32 reference."""
    if name not in _ORACLE_CACHE:
        params, batch, hard, dv, dq, nv = _size_case(name)
        bt = {k: (v.float() if torch.is_tensor(v) else v) for k, v in batch.items()}
        p = {k: v.float() for k, v in params.items()}
        mom = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in p.items()}
        losses, grads0 = [], None
        nthr = torch.get_num_threads()
        torch.set_num_threads(min(nthr, 32))              # 256 threads on the bench boxes: fork / join costs more than it buys
        try:
            for step in range(N_STEPS):
                torch.manual_seed(SEED0 + step)
                rnd = [orc.draw_triplet_randoms(batch["text_labels"], nv, hard, 20) for _ in range(2)]
                pg = {k: v.clone().requires_grad_(True) for k, v in p.items()}
                with torch.enable_grad():
                    d = orc.forward_losses(pg, bt, _cfg(hard, 1.0), rnd)
                    d["loss"].backward()
                losses.append({k: float(v.detach()) for k, v in d.items()})
                if step == 0:
                    grads0 = {k: v.grad.detach().clone() for k, v in pg.items()}
                for k in p:
                    wd = 0.0 if any(nd in k for nd in NO_DECAY) else 0.01
                    p[k], m1, v1 = orc.bert_adam_step(p[k], pg[k].grad, mom[k][0], mom[k][1], step, LR, wd, STEPS_PER_EPOCH * N_EPOCH,
                                                      WARMUP)
                    mom[k] = (m1, v1)
        finally:
            torch.set_num_threads(nthr)
        _ORACLE_CACHE[name] = (losses, grads0, p)
    return _ORACLE_CACHE[name]


@pytest.mark.parametrize("prec", ["fp32", "bf16", "mixed"])
@pytest.mark.parametrize("name", ["c3", "c5"])
def test_train_step_at_full_size_train_mode_vs_oracle(monkeypatch, name, prec):
    """BASELINE configs[2] / configs[4] size, model.train(), dropout 0: the 7 losses AND all 74 gradients of the train-mode path
    (padding skipped) against the oracle's autograd (fp32: see _oracle_trajectory).  Parity mode: losses 1e-4 (north_star),
    gradients 1e-3 of the tensor's norm (measured 5e-6 at C5); throughput mode: losses 2e-2, gradients at bf16 grade (operand
    rounding through the stacked GEMMs: measured 0.13 worst / 0.05 median relative l2, cosine >= 0.991)."""
    from dldkd_amd import ops
    params, batch, hard, dv, dq, nv = _size_case(name)
    ref_losses, ref_grads, _ = _oracle_trajectory(name)
    ref = ref_losses[0]
    m = _train_model(dv, dq, params, hard)
    m.weight = 1.0
    dbatch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    log = _FlagLog(monkeypatch)
    ops.set_gemm_precision(prec)
    try:
        torch.manual_seed(SEED0)
        loss, d = m(dbatch)
        m.zero_grad()
        loss.backward()
    finally:
        ops.set_gemm_precision("fp32")
    L = batch["student_videos"].shape[1]
    skipped, groups = log.assert_skipped(batch["student_videos_mask"].sum(1).long().numpy(), L)
    # "mixed" (fp32-grade forward, bf16 backward): the LOSSES are the parity mode's - north_star's 1e-4 - the gradients bf16-grade
    exact_fwd = prec in ("fp32", "mixed")
    tol = 1e-4 if exact_fwd else 2e-2
    worst_loss = 0.0
    for k in LOSS_KEYS + ("loss",):
        got = float(loss if k == "loss" else d[k])
        worst_loss = max(worst_loss, abs(got - ref[k]) / max(abs(ref[k]), 1e-3 if exact_fwd else 0.1))
        assert abs(got - ref[k]) <= tol * max(abs(ref[k]), 1e-3 if exact_fwd else 0.1), (k, got, ref[k])
    print(f"  {name} {prec}: worst loss error {worst_loss:.3e} (relative)")
    if prec == "mixed":
        # the forward GEMMs of "mixed" run two bf16 planes per operand (ops.MIXED_FORWARD = "fp32x2", ~2^-16 per product): the
        # seven losses must sit well inside north_star's 1e-4 (<= 3e-5)
        assert worst_loss <= 3e-5, worst_loss
    stats = _grad_stats(m, ref_grads)
    if prec == "fp32":
        _assert_grads(stats, 1e-3, f"{name} parity ({skipped}/{groups} groups skipped)")
    elif prec == "mixed":
        _assert_grads(stats, GRAD_TOL_MIXED, f"{name} mixed ({skipped}/{groups} groups skipped)", cos_min=0.97)
    else:
        _assert_grads(stats, GRAD_TOL_BF16, f"{name} bf16 ({skipped}/{groups} groups skipped)", cos_min=0.97)


@pytest.mark.parametrize("prec", ["fp32", "mixed"])
def test_padded_query_axis_at_c5_size_vs_oracle(prec):
    """BASELINE configs[4] (one rank's Charades step: 257 queries) with the query axis padded to the stepper's bucket (288 rows: what
    train.GraphedTrainStep feeds the model once a run's caption counts vary) against the ORACLE on the raw batch: the 7 losses within
    north_star's 1e-4 and all 74 gradients at the mode's tolerance - the padding queries change nothing the reference computes."""
    from dldkd_amd import ops
    from dldkd_amd import train as T
    from dldkd_amd.optimization import BertAdam
    params, batch, hard, dv, dq, nv = _size_case("c5")
    ref_losses, ref_grads, _ = _oracle_trajectory("c5")
    ref = ref_losses[0]
    m = _train_model(dv, dq, params, hard)
    m.weight = 1.0
    dbatch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    stepper = T.GraphedTrainStep(m, BertAdam([{"params": list(m.parameters()), "weight_decay": 0.01}], lr=1e-3, warmup=0.1, t_total=40),
                                 types.SimpleNamespace(grad_clip=-1))
    stepper.vary_queries = True
    padded = stepper._bucketed(dbatch)
    assert padded["student_text"].shape[0] == 288 and len(padded["text_labels"]) == 257
    ops.set_gemm_precision(prec)
    try:
        torch.manual_seed(SEED0)
        loss, d = m(padded)
        m.zero_grad()
        loss.backward()
    finally:
        ops.set_gemm_precision("fp32")
    worst = 0.0
    for k in LOSS_KEYS + ("loss",):
        got = float(loss if k == "loss" else d[k])
        worst = max(worst, abs(got - ref[k]) / max(abs(ref[k]), 1e-3))
    print(f"  c5 padded {prec}: worst loss error {worst:.3e} (relative)")
    assert worst <= (1e-4 if prec == "fp32" else 3e-5), worst
    stats = _grad_stats(m, ref_grads)
    _assert_grads(stats, 1e-3 if prec == "fp32" else GRAD_TOL_MIXED, f"c5 padded {prec}", cos_min=None if prec == "fp32" else 0.97)


# --------------------------------------------------------------------------- the replayed multi-graph step + fused BertAdam
@pytest.mark.parametrize("name,prec", [("c5", "fp32"), ("c5", "bf16"), ("c3", "fp32"), ("c3", "bf16"), ("c5", "mixed"), ("c3", "mixed")])
def test_replayed_multi_graph_step_and_bert_adam_vs_oracle(name, prec):
    """What train() runs: train.GraphedTrainStep with the towers as parallel graphs (one GPU) + the fused BertAdam update, three
    steps on one batch (step 0 eager, step 1 captured and replayed, step 2 replayed), model.train(), dropout 0.  Every step's 7
    losses against the oracle's trajectory (oracle forward / autograd / BertAdam, the reference's weight-decay groups and
    warm-up schedule) and the parameters after the last step against the oracle's."""
    from dldkd_amd import ops
    from dldkd_amd import train as T
    params, batch, hard, dv, dq, nv = _size_case(name)
    n_steps, lr, warmup, spe, n_epoch = N_STEPS, LR, WARMUP, STEPS_PER_EPOCH, N_EPOCH
    ref_losses, _, ref_params = _oracle_trajectory(name)
    m = _train_model(dv, dq, params, hard)
    m.weight = 1.0
    topt = types.SimpleNamespace(grad_clip=-1, lr=lr, wd=0.01, lr_warmup_proportion=warmup, n_epoch=n_epoch)
    dbatch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    ops.set_gemm_precision(prec)
    try:
        optim = T.make_optimizer(m, topt, spe)
        stepper = T.GraphedTrainStep(m, optim, topt)
        tol = 1e-4 if prec == "fp32" else 2e-2       # ("mixed": step 0's losses are parity-grade; later steps run on parameters
        for step in range(n_steps):                   #  moved by bf16-grade gradients: gated like the throughput mode there)
            torch.manual_seed(SEED0 + step)
            loss, d = stepper(dbatch)
            for k in LOSS_KEYS + ("loss",):
                got = float(loss if k == "loss" else d[k])
                ref = ref_losses[step][k]
                # the oracle's parameters drift from ours by the steps' rounding: the later steps get twice the tolerance
                if prec == "mixed" and step == 0:
                    assert abs(got - ref) <= 1e-4 * max(abs(ref), 1e-3), (step, k, got, ref)
                assert abs(got - ref) <= (1 if step < 2 else 2) * tol * max(abs(ref), 1e-3 if prec == "fp32" else 0.1), (step, k, got, ref)
        torch.cuda.synchronize()
    finally:
        ops.set_gemm_precision("fp32")
    assert stepper.eager_steps == 1 and stepper.captures == 1 and stepper.replays == n_steps - 1
    e = next(iter(stepper.graphs.values()))
    assert getattr(e, "par", None), "one GPU: the step must have been captured as parallel tower graphs"
    assert optim.step_count == n_steps
    # parameters: the update of a step is lr_t * (m / (sqrt(v) + eps) + wd p), |m / sqrt(v)| <= ~3.2 -> compare the MOVEMENT
    p0 = {k: v.double() for k, v in params.items()}
    worst, tot_err, tot_mov = 0.0, 0.0, 0.0
    frac_bad, n_el = 0, 0
    big = max(float((ref_params[n].double() - p0[n]).abs().max()) for n in p0)
    for n, prm in m.named_parameters():
        mov_ref = (ref_params[n].double() - p0[n]).reshape(-1)
        mov = (prm.detach().double().cpu() - p0[n]).reshape(-1)
        scale = max(float(mov_ref.abs().max()), 1e-3 * big)      # key biases: zero gradient, no weight decay - they do not move
        err = (mov - mov_ref).abs()
        worst = max(worst, float(err.max()) / scale)
        tot_err += float(err.sum())
        tot_mov += float(mov_ref.abs().sum())
        frac_bad += int((err > 0.05 * scale).sum())
        n_el += err.numel()
    print(f"  {name} {prec}: parameter movement after {n_steps} steps: mean |err| / mean |move| = {tot_err / tot_mov:.3e}, "
          f"worst element {worst:.3e} of its tensor's largest move, {frac_bad} of {n_el} elements off by > 5 %")
    if prec == "fp32":
        assert tot_err / tot_mov <= 1e-3 and frac_bad <= 1e-4 * n_el, (tot_err / tot_mov, frac_bad)
    else:
        assert tot_err / tot_mov <= 0.15, tot_err / tot_mov      # measured 0.07 - 0.08


# ------------------------------------------------------------------------------------------------ "mixed" precision with dropout on
def test_mixed_mode_with_dropout_draws_the_parity_modes_masks_and_agrees_with_it():
    """ops.set_gemm_precision("mixed"): fp32-grade forward chain + the fused bf16 backward (functional._TowerTrainMixed).  With
    dropout ON the oracle cannot be the yardstick (its masks are torch's), but the parity mode can: same seed -> the same Philox
    slots in the same order -> the same masks in every layer (input projection, position LayerNorm, attention probabilities, hidden)
    - so the seven losses agree to 1e-5 and the gradients to bf16-product grade.  A mask convention that differed anywhere (e.g. the
    fp32 attention forward against the bf16 attention backward's recomputed bits) would show as a gradient error of order one."""
    from dldkd_amd import ops
    params, batch, hard, dv, dq, nv = _size_case("c5")
    dbatch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    res = {}
    try:
        for prec in ("fp32", "mixed"):
            ops.set_gemm_precision(prec)
            m = _train_model(dv, dq, params, hard, drop=0.15)
            m.weight = 1.0
            torch.manual_seed(SEED0)
            loss, d = m(dbatch)
            m.zero_grad()
            loss.backward()
            torch.cuda.synchronize()
            res[prec] = ({k: float(d[k]) for k in LOSS_KEYS}, {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()})
    finally:
        ops.set_gemm_precision("fp32")
    for k in LOSS_KEYS:
        a, b = res["mixed"][0][k], res["fp32"][0][k]
        assert abs(a - b) <= 1e-5 * max(abs(b), 1e-3), (k, a, b)
    nmax = max(float(g.norm()) for g in res["fp32"][1].values())
    worst = 0.0
    for n, r in res["fp32"][1].items():
        g = res["mixed"][1][n]
        rel = float((g - r).norm()) / max(float(r.norm()), 1e-4 * nmax)
        worst = max(worst, rel)
        assert rel <= 0.03, (n, rel)
    print(f"  mixed vs parity with dropout 0.15 (same masks): worst gradient rel l2 difference {worst:.3e}")
