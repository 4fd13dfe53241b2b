"""Timing of the fused per-sequence tower kernel K5 (tower_seq.hip) on synthetic TVR-shaped rows: HIP events around the kernel
alone, interleaved variants in one process (full-length vs ragged, one sequence per workgroup vs packed slots)."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dl-dkd_amd"))
import numpy as np
import torch


def main(n=1024, iters=12, dev="cuda:0"):
    from dldkd_amd.model import DLDKD
    from dldkd_amd import ops, scoring
    cfg = types.SimpleNamespace(visual_input_size=3072, query_input_size=768, inheritance_hidden=384, exploration_hidden=384,
                                max_ctx_l=128, max_desc_l=30, input_drop=0.2, drop=0.2, n_heads=4, initializer_range=0.02,
                                margin=0.1, use_hard_negative=True, hard_pool_size=20, label_style="soft")
    opt_ = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                 collection="tvr", alpha=0.8, belta=0.8)
    torch.manual_seed(0)
    m = DLDKD(cfg, opt_).to(dev).eval()
    ops.set_gemm_precision("bf16")
    packs = m._tower_packs("visual")
    g = torch.Generator(device=dev).manual_seed(1)
    h0 = [torch.relu(torch.randn(n, 128, 384, generator=g, device=dev)) for _ in range(2)]
    lens_full = torch.full((n,), 128, dtype=torch.int32, device=dev)
    lens_rag = torch.randint(24, 129, (n,), generator=g, device=dev).to(torch.int32)
    items_rag = torch.from_numpy(ops.plan_tower_items(lens_rag.cpu().numpy())).to(dev)
    pk = scoring.GalleryPacker(n, 128, 2, torch.device(dev))
    variants = {
        "full128_rows": dict(lens=lens_full, items=None, out_mode=0),
        "full128_gallery": dict(lens=lens_full, items=None, out_mode=1),
        "ragged_gallery_1seq_per_wg": dict(lens=lens_rag, items=None, out_mode=1),
        "ragged_gallery_packed": dict(lens=lens_rag, items=items_rag, out_mode=1),
    }
    times = {k: [] for k in variants}

    def run(v):
        if v["out_mode"] == 1:
            ops.tower_seq(h0, packs, v["lens"], seq_rows=128, items=v["items"], out_mode=1, gallery=pk.blobs, v0=0, Lp=pk.Lp, lens_out=pk.lens)
        else:
            ops.tower_seq(h0, packs, v["lens"], seq_rows=128, items=v["items"])
    with torch.no_grad():
        for v in variants.values():
            run(v)
        torch.cuda.synchronize()
        for _ in range(iters):
            for k, v in variants.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run(v)
                e1.record()
                e1.synchronize()
                times[k].append(e0.elapsed_time(e1))
    ops.set_gemm_precision("fp32")
    out = {"n_sequences": n, "branches": 2}
    flops_seq = 2 * 128 * 384 * (1152 + 384 + 384) + 4 * 2 * 2 * 128 * 128 * 96
    for k, t in times.items():
        ms = float(np.median(t))
        rows = float(variants[k]["lens"].sum().item())
        out[k] = {"ms_median": ms, "ms_min": float(min(t)), "videos_per_s": n / ms * 1e3,
                  "TFLOPs_at_padded_128": 2 * n * flops_seq / ms / 1e9, "valid_rows": rows,
                  "workgroups": int(variants[k]["items"].shape[0] if variants[k]["items"] is not None else n) * 2}
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1024)
