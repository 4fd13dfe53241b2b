"""The R@K gate at TVR dimensions (Dv = 3072 i3d clips, Dq = 768 RoBERTa words: /root/reference/do_tvr.sh:5-16) with a TRAINED model.

tools/rk_gate.py plants its signal by giving the query towers the video towers' weights, which forces Dv = Dq (ActivityNet dims) and
leaves the K = 3072 projection - three times the bf16 accumulation length - outside the gate.  Here the model becomes
non-chance the way the reference's does: a few hundred steps of this repo's own parity-mode training loop (dldkd_amd.train:
GraphedTrainStep + fused BertAdam) on planted synthetic pairs - a caption's words are noisy images of one clip of its video under a
fixed random map R^3072 -> R^768, the teacher's 512-d features under another - then eval sets the model never saw are scored three
ways from RAW features:
    oracle   fp32 oracle towers + oracle scoring on the CPU (the reference's arithmetic)
    parity   HIP, fp32-grade towers, bf16 scorer
    fast     HIP throughput mode (bf16 input projection, fused bf16 tower kernel, bf16 scorer) on padded fp32 super-batches
    resident the same from the ragged bf16 feature table (K4b -> bf16 h0 rows -> fused tower): what eval_epoch runs by default
For every eval seed: R@1/5/10/100 of each, the NET deltas against the oracle and the GROSS number of queries that cross each cut
in either direction (a net delta can hide crossings that cancel).

    python tools/rk_gate_tvr.py [--seeds 3] [--nv 4096] [--nq 8192] [--steps 400] [--out profiles/r04/rk_gate.json]
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (os.path.join(ROOT, "dl-dkd_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)
import torch
import torch.nn.functional as F

DV, DQ, DT = 3072, 768, 512


def maps(seed=1234):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(DV, DQ, generator=g) / DV ** 0.5, torch.randn(DV, DT, generator=g) / DV ** 0.5


def make_pairs(seed, nv, caps, L, len_lo, sigma, P, Pt, lq_lo=5, lq_hi=30, dev="cpu"):
    """nv videos with `caps` captions each.  Returns the eval-style dict (vid, vmask, lens, words, qmask, gt) + teacher features.
    dev: where the tensors are generated (training batches: on the GPU - 128 x 64 x 3072 normals per step cost 0.5 s on the host)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    P, Pt = P.to(dev), Pt.to(dev)
    lens = torch.randint(len_lo, L + 1, (nv,), generator=g, device=dev)
    lens[0] = L
    vmask = (torch.arange(L, device=dev)[None, :] < lens[:, None]).float()
    vid = F.normalize(torch.randn(nv, L, DV, generator=g, device=dev), dim=-1) * vmask[..., None]
    nq = nv * caps
    gt = torch.arange(nq, device=dev) // caps
    qlens = torch.randint(lq_lo, lq_hi + 1, (nq,), generator=g, device=dev)
    qmask = (torch.arange(lq_hi, device=dev)[None, :] < qlens[:, None]).float()
    clip = (torch.rand(nq, generator=g, device=dev) * lens[gt]).long()
    base = vid[gt, clip]                                                     # (nq, DV): the clip a caption describes
    words = (base @ P)[:, None, :] + sigma / DQ ** 0.5 * torch.randn(nq, lq_hi, DQ, generator=g, device=dev)
    words = F.normalize(words, dim=-1) * qmask[..., None]
    return dict(vid=vid, vmask=vmask, lens=lens, words=words, qmask=qmask, gt=gt, t_vid=3.0 * (vid @ Pt) * vmask[..., None],
                t_txt=3.0 * (base @ Pt)[:, None, :])


def train_model(steps, sigma, P, Pt, dev="cuda:0", seed=0, bsz=128, caps=5, L=64, log=None, precision="fp32", every=100):
    """Training on planted pairs: `steps` steps of 128 videos x 5 captions through train.GraphedTrainStep.  precision "fp32" = parity
    mode (fp32-grade GEMMs), "bf16" = the throughput training mode.  Same seed -> same initial weights, same batches and the same
    triplet draws in either mode.  Returns (model in eval mode, [(step, loss)] every `every` steps)."""
    from dldkd_amd import ops
    from dldkd_amd import train as T
    from dldkd_amd.model import DLDKD
    cfg = types.SimpleNamespace(visual_input_size=DV, query_input_size=DQ, inheritance_hidden=384, exploration_hidden=384, max_ctx_l=128,
                                max_desc_l=30, input_drop=0.1, drop=0.1, n_heads=4, initializer_range=0.02, margin=0.1,
                                use_hard_negative=True, hard_pool_size=20, label_style="soft")
    mopt = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04, collection="tvr",
                                 alpha=0.8, belta=0.8)
    torch.manual_seed(seed)
    m = DLDKD(cfg, mopt).to(dev).train()
    topt = types.SimpleNamespace(grad_clip=-1, lr=3e-4, wd=0.01, lr_warmup_proportion=0.01, n_epoch=1)
    ops.set_gemm_precision(precision)
    optim = T.make_optimizer(m, topt, steps)
    stepper = T.GraphedTrainStep(m, optim, topt, defer_loss_float=True)
    losses = []
    labels = [i // caps for i in range(bsz * caps)]
    for it in range(steps):
        d = make_pairs(10_000 + it, bsz, caps, L, 8, sigma, P, Pt, dev=dev)
        batch = {"student_videos": d["vid"], "student_videos_mask": d["vmask"], "teacher_videos": d["t_vid"], "student_text": d["words"],
                 "student_text_mask": d["qmask"], "teacher_text": d["t_txt"], "text_labels": labels}
        loss, parts = stepper(batch)
        if it % every == 0 or it == steps - 1:
            losses.append((it, float(loss)))
            if log and (it % 100 == 0 or it == steps - 1):
                log(f"  train step {it}: loss {float(loss):.4f}  " + " ".join(f"{k} {float(v):.3f}" for k, v in parts.items() if k != "loss_overall"))
    ops.set_gemm_precision("fp32")
    return m.eval(), losses


def compare(m, d, modes=("parity", "fast", "resident"), dev="cuda:0", chunk=50):
    import rk_gate
    out = {}
    ref, _, _ = rk_gate.oracle_scores(m, d, threads=32, chunk=chunk)
    out["oracle"], r_ref = rk_gate.recalls(ref, d["gt"])
    for mode in modes:
        fused, _, _ = rk_gate.hip_scores(m, d, mode, dev)
        rk, r = rk_gate.recalls(fused, d["gt"])
        out[mode] = {"recall": rk, "delta_vs_oracle": [a - b for a, b in zip(rk, out["oracle"])],
                     "crossings_pct": [100.0 * float(((r <= k) != (r_ref <= k)).mean()) for k in (1, 5, 10, 100)],
                     "queries_crossing_a_cut": [int(((r <= k) != (r_ref <= k)).sum()) for k in (1, 5, 10, 100)],
                     "queries_whose_rank_changed": int((r != r_ref).sum()),
                     "max_abs_score_err": float((fused - ref).abs().max()), "mean_abs_score_err": float((fused - ref).abs().mean())}
    return out


def run(seeds=3, nv=4096, nq=8192, steps=400, sigma=6.0, L=64, log=print, chunk=50):
    P, Pt = maps()
    t0 = time.time()
    m, losses = train_model(steps, sigma, P, Pt, log=log)
    res = {"dims": {"Dv": DV, "Dq": DQ}, "train": {"steps": steps, "batch_videos": 128, "captions_per_video": 5, "loss": losses,
                                                       "seconds": round(time.time() - t0, 1), "precision": "parity (fp32-grade GEMMs)"},
           "eval": {"n_videos": nv, "n_queries": nq, "max_clips": L, "sigma": sigma, "one_query_pct": 100.0 / nq}, "seeds": []}
    caps = nq // nv
    for s in range(seeds):
        d = {k: v.cpu() for k, v in make_pairs(500 + s, nv, caps, L, 8, sigma, P, Pt, dev="cuda:0").items()}
        t1 = time.time()
        r = compare(m, d, chunk=chunk)
        r["seed"], r["seconds"] = 500 + s, round(time.time() - t1, 1)
        res["seeds"].append(r)
        log(f"  eval seed {500 + s}: oracle {['%.3f' % x for x in r['oracle']]}  " + "  ".join(
            f"{k} d {['%+.3f' % x for x in r[k]['delta_vs_oracle']]} x {r[k]['queries_crossing_a_cut']}" for k in ("parity", "fast", "resident")))
    for mode in ("parity", "fast", "resident"):
        res[mode + "_worst_abs_net_delta"] = max(abs(x) for r in res["seeds"] for x in r[mode]["delta_vs_oracle"])
        res[mode + "_worst_gross_crossings_pct"] = max(x for r in res["seeds"] for x in r[mode]["crossings_pct"])
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--nv", type=int, default=4096)
    ap.add_argument("--nq", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--sigma", type=float, default=6.0)
    ap.add_argument("--out", default="")
    ap.add_argument("--chunk", type=int, default=512, help="queries per oracle get_sim_scores call (50 = the reference's eval_query_bsz; same arithmetic)")
    a = ap.parse_args()
    res = run(a.seeds, a.nv, a.nq, a.steps, a.sigma, chunk=a.chunk)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "seeds"}))
