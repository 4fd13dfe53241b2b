"""R@K of the eval path from RAW features, three ways on the same planted-signal data: the fp32 oracle towers + oracle scoring
(CPU), the HIP parity mode (fp32-grade towers, bf16 scorer) and the HIP throughput mode (K4 + fused bf16 tower kernel + bf16
scorer).  north_star gates eval numbers at R@1/5/10/100 within +-0.1 of the reference; this tool measures the distance and says
which stage moves ranks (tests/test_rk_gate_gpu.py asserts on its output).

Signal is planted in FEATURE space: with Dv = Dq the query towers are given the video towers' weights (and the out mapping is
the identity), a query's words are noisy copies of one clip of its ground-truth video; sigma sets R@1 (15-40 % like TVR)."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (os.path.join(ROOT, "dl-dkd_amd"), os.path.join(ROOT, "oracle")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)
import numpy as np
import torch
import torch.nn.functional as F


def make_model(D=1024, seed=0, dev="cuda:0"):
    from dldkd_amd.model import DLDKD
    cfg = types.SimpleNamespace(visual_input_size=D, query_input_size=D, inheritance_hidden=384, exploration_hidden=384,
                                max_ctx_l=128, max_desc_l=30, input_drop=0.2, drop=0.2, n_heads=4, initializer_range=0.02,
                                margin=0.1, use_hard_negative=False, hard_pool_size=20, label_style="soft")
    opt = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                collection="anet", alpha=0.8, belta=0.8)
    torch.manual_seed(seed)
    m = DLDKD(cfg, opt)
    sd = m.state_dict()
    for pre in ("", "exp_"):                      # query towers := video towers, out mapping := identity
        for k in list(sd):
            if k.startswith(pre + "visual_input_proj.") or k.startswith(pre + "visual_encoder."):
                sd[k.replace("visual_", "query_", 1)] = sd[k].clone()
        sd[pre + "query_pos_embed.LayerNorm.weight"] = sd[pre + "visual_pos_embed.LayerNorm.weight"].clone()
        sd[pre + "query_pos_embed.LayerNorm.bias"] = sd[pre + "visual_pos_embed.LayerNorm.bias"].clone()
        sd[pre + "query_pos_embed.position_embeddings.weight"] = sd[pre + "visual_pos_embed.position_embeddings.weight"][:30].clone()
        sd[pre + "out_mapping_linear.weight"] = torch.eye(384)
        sd[pre + "out_mapping_linear.bias"] = torch.zeros(384)
    m.load_state_dict(sd)
    return m.to(dev).eval()


def make_data(seed, nv, nq, L=128, D=1024, sigma=1.0, len_lo=16):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(len_lo, L + 1, (nv,), generator=g)
    lens[0] = L
    vmask = (torch.arange(L)[None, :] < lens[:, None]).float()
    vid = F.normalize(torch.randn(nv, L, D, generator=g), dim=-1) * vmask[..., None]
    qlens = torch.randint(5, 31, (nq,), generator=g)
    qmask = (torch.arange(30)[None, :] < qlens[:, None]).float()
    gt = torch.arange(nq) % nv
    clip = (torch.rand(nq, generator=g) * lens[gt]).long()
    base = vid[gt, clip]
    words = base[:, None, :] + sigma / D ** 0.5 * torch.randn(nq, 30, D, generator=g)
    words = F.normalize(words, dim=-1) * qmask[..., None]
    return dict(vid=vid, vmask=vmask, lens=lens, words=words, qmask=qmask, gt=gt)


def recalls(scores, gt):
    s = scores.double()
    gs = s.gather(1, gt.view(-1, 1).to(s.device))
    rank = 1 + (s > gs).sum(1)
    r = rank.cpu().numpy()
    return [100.0 * float((r <= k).mean()) for k in (1, 5, 10, 100)], r


def oracle_scores(m, d, threads=16, chunk=50):
    """chunk: queries per get_sim_scores call (the reference's eval_query_bsz = 50, eval.py:188-208; a larger chunk is the same
    arithmetic - every call re-normalises the gallery and takes the same fp32 products - with fewer passes over the gallery)."""
    import dldkd_oracle as orc
    torch.set_num_threads(min(threads, os.cpu_count() or 1))
    p = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        gi, ge = [], []
        for s in range(0, d["vid"].shape[0], 256):
            a, b = orc.encode_context(p, d["vid"][s:s + 256], d["vmask"][s:s + 256])
            gi.append(a), ge.append(b)
        gi, ge = torch.cat(gi), torch.cat(ge)
        qi, qe = orc.encode_query(p, d["words"], d["qmask"])
        inh, exp = orc.eval_scores(qi, qe, gi, ge, d["vmask"], chunk=chunk)
    return orc.fuse_scores(inh, exp), inh, exp


def hip_scores(m, d, mode, dev="cuda:0", chunk=512, tower_seq=True):
    """mode 'parity' | 'fast' (K4 + fused tower) | 'resident' (eval_epoch's default: ragged bf16 feature table -> K4b -> bf16 h0 ->
    fused tower) | 'fast_chain' (the kernel chain K5 replaced) | 'k4_only' (bf16 K4, fp32 towers)"""
    from dldkd_amd import ops, scoring
    ops.set_gemm_precision("bf16" if mode in ("fast", "fast_chain", "resident") else "fp32")
    m.fast_input_proj = mode != "parity"
    ops.TOWER_SEQ = mode != "fast_chain"
    try:
        nv, L = d["vid"].shape[:2]
        pk = scoring.GalleryPacker(nv, L, 2, torch.device(dev))
        with torch.no_grad():
            if mode == "resident":
                from dldkd_amd import eval as ev
                m.eval()
                assert m.resident_encode_ok()
                res = ev.ResidentGallery(d["vid"].shape[2], torch.device(dev))
                for s in range(0, nv, chunk):
                    res.table.append(d["vid"][s:s + chunk].to(dev), d["lens"][s:s + chunk].numpy())
                res.complete = True
                res.plan(torch.device(dev))
                m.encode_resident_into(pk, res)
            for s in range(0, nv if mode != "resident" else 0, chunk):
                v, vm = d["vid"][s:s + chunk].to(dev), d["vmask"][s:s + chunk].to(dev)
                if not (mode == "fast" and m.encode_context_into(pk, v, vm, lens_host=d["lens"][s:s + chunk].numpy())):
                    gi, ge = m.encode_context(v, vm)
                    pk.add([gi, ge], vm)
            pg = pk.finish()
            qi, qe = [], []
            for s in range(0, d["words"].shape[0], 2048):
                a, b = m.encode_query(d["words"][s:s + 2048].to(dev), d["qmask"][s:s + 2048].to(dev))
                qi.append(a), qe.append(b)
            fused, s0, s1 = m.pooled_scores([torch.cat(qi), torch.cat(qe)], pg, want_branches=True)
        return fused.cpu(), s0.cpu(), s1.cpu()
    finally:
        ops.set_gemm_precision("fp32")
        m.fast_input_proj = False
        ops.TOWER_SEQ = True


def run(nv=1536, nq=2048, sigma=1.0, seed=11, modes=("parity", "fast", "fast_chain", "k4_only"), dev="cuda:0"):
    m = make_model(dev=dev)
    d = make_data(seed, nv, nq, sigma=sigma)
    out = {"n_videos": nv, "n_queries": nq, "sigma": sigma, "one_query_pct": 100.0 / nq}
    ref, _, _ = oracle_scores(m, d)
    out["oracle"], r_ref = recalls(ref, d["gt"])
    for mode in modes:
        fused, _, _ = hip_scores(m, d, mode, dev)
        rk, r = recalls(fused, d["gt"])
        out[mode] = {"recall": rk, "delta_vs_oracle": [a - b for a, b in zip(rk, out["oracle"])],
                     "max_abs_score_err": float((fused - ref).abs().max()), "mean_abs_score_err": float((fused - ref).abs().mean()),
                     "queries_whose_rank_changed": int((r != r_ref).sum()),
                     "queries_crossing_a_cut": [int(((r <= k) != (r_ref <= k)).sum()) for k in (1, 5, 10, 100)]}
    return out


if __name__ == "__main__":
    sig = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    nv = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
    nq = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    print(json.dumps(run(nv, nq, sig)))
