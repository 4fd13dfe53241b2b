"""Where a workgroup of the fused tower kernel K5 spends its cycles: s_memtime stamps at the phase boundaries of the gallery-mode
kernel (dldkd_debug_tower_seq_timeline; a diagnostic build of the same code: read the SHARES, not the total)."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dl-dkd_amd"))
import numpy as np
import torch


def main(n=1024, dev="cuda:0", ragged=True, h16=False):
    from dldkd_amd.model import DLDKD
    from dldkd_amd import native, ops, scoring
    cfg = types.SimpleNamespace(visual_input_size=3072, query_input_size=768, inheritance_hidden=384, exploration_hidden=384,
                                max_ctx_l=128, max_desc_l=30, input_drop=0.2, drop=0.2, n_heads=4, initializer_range=0.02,
                                margin=0.1, use_hard_negative=True, hard_pool_size=20, label_style="soft")
    opt_ = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                 collection="tvr", alpha=0.8, belta=0.8)
    torch.manual_seed(0)
    m = DLDKD(cfg, opt_).to(dev).eval()
    packs = [p.get() for p in m._tower_packs("visual")]
    g = torch.Generator(device=dev).manual_seed(1)
    h0 = [torch.relu(torch.randn(n, 128, 384, generator=g, device=dev)) for _ in range(2)]
    lens = (torch.randint(24, 129, (n,), generator=g, device=dev) if ragged else torch.full((n,), 128, device=dev)).to(torch.int32)
    items = torch.from_numpy(ops.plan_tower_items(lens.cpu().numpy())).to(dev)
    pk = scoring.GalleryPacker(n, 128, 2, torch.device(dev))
    n_wg = 8 * ((items.shape[0] + 3) // 4)
    stamps = torch.zeros(n_wg, 24, dtype=torch.int64, device=dev)
    L = native.lib()
    hs = [(x.half() if h16 else x).view(-1, 384) for x in h0]
    for _ in range(3):
        native.check(L.dldkd_debug_tower_seq_timeline(native.ptr_array(hs),
                                                      native.ptr_array([p.blob for p in packs]), native.ptr(lens), native.ptr(items),
                                                      items.shape[0], n, 128, native.ptr_array(pk.blobs), pk.Lp, native.ptr(stamps),
                                                      int(h16), native.stream()), "timeline")
    torch.cuda.synchronize()
    t = stamps.cpu().numpy().astype(np.float64)
    t = t[t[:, 13] > 0]
    d = np.diff(t[:, :14], axis=1)
    names = ["prologue (params, h0 + pos, LN1)"] + [x for h in range(4) for x in (f"head {h}: q|k|v projection (216 MFMAs)", f"head {h}: K/V exchange + attention (48 MFMAs)")] + \
            ["dense (288 MFMAs) + residual", "LayerNorm 2 + pack", "out mapping (288 MFMAs)", "normalise + stage + store rows"]
    tot = (t[:, 13] - t[:, 0])
    out = {"workgroups": int(len(t)), "cycles_per_workgroup_median": float(np.median(tot)), "mfma_floor_cycles": 1632 * 32,
           "phases_median_cycles": {nm: float(np.median(d[:, i])) for i, nm in enumerate(names)}}
    grouped = {"prologue": d[:, 0], "projections": d[:, [1, 3, 5, 7]].sum(1), "attention": d[:, [2, 4, 6, 8]].sum(1), "dense": d[:, 9],
               "layernorm2": d[:, 10], "out_mapping": d[:, 11], "store": d[:, 12]}
    pd = np.diff(np.concatenate([t[:, 0:1], t[:, 16:22], t[:, 1:2]], axis=1), axis=1)
    out["prologue_detail_median_cycles"] = dict(zip(["start -> half-0 DMA issued", "-> half 0 landed (vmcnt 0)", "-> half 0 read + added", "-> half 1 landed",
                                                     "-> half 1 read", "-> workgroup barrier", "-> LN1 applied, chunks 1-2 issued"],
                                                    [float(np.median(pd[:, i])) for i in range(7)]))
    out["share_of_workgroup_time"] = {k: float(np.median(v / tot)) for k, v in grouped.items()}
    # wave 0's cycles parked at the 44 chunk hand-overs of the weight stream: waiting for its own LDS-DMA pieces, then at the barrier
    out["chunk_wait_median_cycles"] = {"own_dma_pieces": float(np.median(t[:, 14])), "barrier": float(np.median(t[:, 15])),
                                       "share_of_workgroup_time": float(np.median((t[:, 14] + t[:, 15]) / tot))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(ragged="full" not in sys.argv[1:], h16="h16" in sys.argv[1:])
