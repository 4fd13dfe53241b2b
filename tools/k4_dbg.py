import os, sys, types
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "dl-dkd_amd"))
import torch
from dldkd_amd import ops
from dldkd_amd.model import DLDKD
K, M = 3072, 1280
cfg = types.SimpleNamespace(visual_input_size=K, query_input_size=768, inheritance_hidden=384, exploration_hidden=384,
                            max_ctx_l=128, max_desc_l=30, input_drop=0.2, drop=0.2, n_heads=4, initializer_range=0.02,
                            margin=0.1, use_hard_negative=True, hard_pool_size=20, label_style="soft")
opt_ = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                             collection="tvr", alpha=0.8, belta=0.8)
torch.manual_seed(0)
m = DLDKD(cfg, opt_).to("cuda:0").eval()
x = torch.nn.functional.normalize(torch.randn(M, K, device="cuda:0"), dim=-1)
f = ops.FoldedInProj([m.visual_input_proj, m.exp_visual_input_proj])
ops.INPROJ_KERNEL = "full"; a = ops.in_proj_bf16(x, f)
ops.INPROJ_KERNEL = "rows128"; b = ops.in_proj_bf16(x, f)
torch.cuda.synchronize()
for br in range(2):
    d = (a[br] - b[br]).abs()
    bad = ~torch.isfinite(b[br]) | (d > 1e-3)
    print("branch", br, "bad", bad.sum().item(), "of", bad.numel())
    rows = bad.any(1).nonzero().flatten()
    cols = bad.any(0).nonzero().flatten()
    print(" bad rows (mod 128) hist by 32:", torch.bincount((rows % 128) // 32, minlength=4).tolist(), " tiles:", torch.unique(rows // 128).tolist()[:20])
    print(" bad cols hist by 32:", torch.bincount(cols // 32, minlength=12).tolist())
    if bad.any():
        r, c = bad.nonzero()[0].tolist()
        print(" first bad", r, c, a[br][r, c].item(), b[br][r, c].item())
