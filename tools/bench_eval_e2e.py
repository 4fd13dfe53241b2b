"""Stage timings of the eval path at TVR scale (GPU box): gallery encode (raw i3d-dim features -> towers ->
packed bf16), query encode, scoring, ranking.  Synthetic features generated on the device in chunks."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dl-dkd_amd"))
import torch
from dldkd_amd.model import DLDKD
from dldkd_amd import scoring, eval as ev
DEV = "cuda:0"
NV = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 10895
cfg = types.SimpleNamespace(visual_input_size=3072, query_input_size=768, inheritance_hidden=384, exploration_hidden=384,
                            max_ctx_l=128, max_desc_l=30, input_drop=0.2, drop=0.2, n_heads=4, initializer_range=0.02,
                            margin=0.1, use_hard_negative=True, hard_pool_size=20, label_style="soft")
opt_ = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                             collection="tvr", alpha=0.8, belta=0.8)
torch.manual_seed(0)
m = DLDKD(cfg, opt_).to(DEV).eval()
gen = torch.Generator(device=DEV).manual_seed(1)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
with torch.no_grad():
    # gallery encode in chunks of 200 videos (eval_context_bsz)
    B, L = 200, 128
    feats = torch.nn.functional.normalize(torch.randn(B, L, 3072, generator=gen, device=DEV), dim=-1)
    lens = torch.randint(24, L + 1, (B,), generator=gen, device=DEV)
    mask = (torch.arange(L, device=DEV).unsqueeze(0) < lens.unsqueeze(1)).float()
    feats = feats * mask.unsqueeze(-1)
    for _ in range(2): m.encode_context(feats, mask)
    t0 = sync(); inh, exp = [], []
    nchunk = (NV + B - 1) // B
    for _ in range(nchunk):
        gi, ge = m.encode_context(feats, mask); inh.append(gi); exp.append(ge)
    t1 = sync()
    gi, ge, mk = torch.cat(inh)[:NV], torch.cat(exp)[:NV], mask.repeat(nchunk, 1)[:NV]
    pg = scoring.pack_gallery([gi, ge], mk)
    t2 = sync()
    print(f"gallery encode (fp32 towers): {nchunk*B} videos in {t1-t0:.3f}s = {nchunk*B/(t1-t0):.0f} videos/s "
          f"({nchunk*B*L*3072*4/(t1-t0)/1e9:.0f} GB/s of raw features); pack {t2-t1:.3f}s")
    words = torch.nn.functional.normalize(torch.randn(50, 30, 768, generator=gen, device=DEV), dim=-1)
    wmask = torch.ones(50, 30, device=DEV)
    for _ in range(2): m.encode_query(words, wmask)
    t0 = sync(); qi, qe = [], []
    for _ in range((NQ + 49) // 50):
        a, b = m.encode_query(words, wmask); qi.append(a); qe.append(b)
    t1 = sync()
    print(f"query encode: {NQ} queries in {t1-t0:.3f}s = {NQ/(t1-t0):.0f} queries/s")
    qs = [torch.cat(qi)[:NQ], torch.cat(qe)[:NQ]]
    m.pooled_scores(qs, pg)
    t0 = sync(); fused, s0, s1 = m.pooled_scores(qs, pg, want_branches=True); t1 = sync()
    print(f"scoring ({NQ} x {NV}, 3 matrices out): {1e3*(t1-t0):.2f} ms")
    gt = {q: [q % NV] for q in range(NQ)}
    ev.gt_ranks_gpu(fused, gt)
    t0 = sync(); rb, rf = ev.gt_ranks_gpu(fused, gt); r = rb.cpu(); t1 = sync()
    print(f"ranking on GPU (incl. CSR build + D2H of {NQ} ranks): {1e3*(t1-t0):.2f} ms")
