import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dl-dkd_amd"))
import torch
from dldkd_amd.model import DLDKD
DEV = "cuda:0"
cfg = types.SimpleNamespace(visual_input_size=3072, query_input_size=768, inheritance_hidden=384, exploration_hidden=384,
                            max_ctx_l=128, max_desc_l=30, input_drop=0.2, drop=0.2, n_heads=4, initializer_range=0.02,
                            margin=0.1, use_hard_negative=True, hard_pool_size=20, label_style="soft")
opt_ = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                             collection="tvr", alpha=0.8, belta=0.8)
m = DLDKD(cfg, opt_).to(DEV).eval()
m.fast_input_proj = len(sys.argv) > 1 and sys.argv[1] in ("fast", "allbf16")
if len(sys.argv) > 1 and sys.argv[1] == "allbf16":
    from dldkd_amd import ops
    ops.set_gemm_precision("bf16")
NB = int(os.environ.get("ENC_BATCH", "200"))
feats = torch.nn.functional.normalize(torch.randn(NB, 128, 3072, device=DEV), dim=-1)
mask = torch.ones(NB, 128, device=DEV)
if len(sys.argv) > 1 and sys.argv[1] == "fused":
    # round 3: what eval_epoch runs in throughput mode - K4 + the fused tower kernel K5 straight into the packed bf16 gallery,
    # ragged lengths U{24..128}, short videos sharing workgroups
    from dldkd_amd import ops, scoring
    ops.set_gemm_precision("bf16")
    m.fast_input_proj = True
    g = torch.Generator(device=DEV).manual_seed(1)
    lens = torch.randint(24, 129, (NB,), generator=g, device=DEV)
    mask = (torch.arange(128, device=DEV)[None] < lens[:, None]).float()
    feats = feats * mask[..., None]
    lh = lens.cpu().numpy()
    pk = scoring.GalleryPacker(NB * 14, 128, 2, torch.device(DEV))
    with torch.no_grad():
        for _ in range(3): m.encode_context_into(pk, feats, mask, lens_host=lh)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): m.encode_context_into(pk, feats, mask, lens_host=lh)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"encode_context_into {NB}x128x3072 ragged U{{24..128}}: {dt*1e3:.2f} ms = {NB/dt:.0f} videos/s")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "resident":
    # round 3, second half: what eval_epoch runs in throughput mode - the whole gallery (NV videos, ragged U{24..128}) resident as a
    # bf16 table with row statistics, ONE K4b launch + ONE fused tower launch per encode
    from dldkd_amd import ops, scoring, eval as ev
    ops.set_gemm_precision("bf16")
    m.fast_input_proj = True
    NV = int(os.environ.get("ENC_VIDEOS", "21793"))
    g = torch.Generator(device=DEV).manual_seed(1)
    lens = torch.randint(24, 129, (NB,), generator=g, device=DEV)
    feats = feats * (torch.arange(128, device=DEV)[None] < lens[:, None]).float()[..., None]
    lh = lens.cpu().numpy()
    res = ev.ResidentGallery(3072, torch.device(DEV))
    done = 0
    while done < NV:
        n = min(NB, NV - done)
        res.table.append(feats[:n], lh[:n])
        done += n
    res.complete = True
    res.plan(torch.device(DEV))
    with torch.no_grad():
        # as eval._resident_context_info does for a cached table: the packed gallery's buffers are zero-filled once and re-encoded in
        # place (DLDKD_SKIP_ZERO_ROWS=0: the kernel writes the padding rows every time)
        first = scoring.GalleryPacker(NV, 128, 2, torch.device(DEV), zero_fill=True)
        mk = lambda: scoring.GalleryPacker(NV, 128, 2, torch.device(DEV), blobs=first.blobs)      # noqa: E731
        for _ in range(2):
            pk = mk(); m.encode_resident_into(pk, res)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            pk = mk(); m.encode_resident_into(pk, res)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"encode_resident_into {NV} videos ({res.table.rows} clips, {res.table.nbytes()/1e9:.1f} GB resident): {dt*1e3:.2f} ms = {NV/dt:.0f} videos/s")
    sys.exit(0)
with torch.no_grad():
    for _ in range(3): m.encode_context(feats, mask)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m.encode_context(feats, mask)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"encode_context {NB}x128x3072 fast={m.fast_input_proj}: {dt*1e3:.2f} ms = {NB/dt:.0f} videos/s")
