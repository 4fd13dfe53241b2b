import os, sys, types
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "dl-dkd_amd"))
import torch
from dldkd_amd import ops
from dldkd_amd.model import DLDKD
K = 3072
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
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
    print("branch", br, "bad", bad.sum().item(), "of", bad.numel(), "max", d.max().item())
    rows = bad.any(1).nonzero().flatten()
    cols = bad.any(0).nonzero().flatten()
    print(" bad rows (mod 128) hist by 32:", torch.bincount((rows % 128) // 32, minlength=4).tolist(), " tiles:", torch.unique(rows // 128).tolist()[:40])
    print(" bad cols hist by 32:", torch.bincount(cols // 32, minlength=12).tolist())
    if bad.any():
        r, c = bad.nonzero()[0].tolist()
        print(" first bad", r, c, a[br][r, c].item(), b[br][r, c].item())
from dldkd_amd import native
L = native.lib()
fo = f.get()
z = torch.zeros_like(fo.cs)
outs = {}
for name, fn in (("full", L.dldkd_in_proj_bf16_full), ("rows128", L.dldkd_in_proj_bf16_rows128)):
    ys = [torch.empty(M, 384, device="cuda:0") for _ in range(2)]
    native.check(fn(native.ptr(x), native.ptr(fo.Wf), native.ptr(z), native.ptr(z), native.ptr(ys[0]), native.ptr(ys[1]), M, K, 1e-5, 0, native.stream()), name)
    outs[name] = ys
torch.cuda.synchronize()
r = outs["rows128"][0] / outs["full"][0]
print("cs=bb=0: ratio rows128/full, row 0 cols 0..5:", r[0, :6].tolist())
print("  per-row std of ratio (first 6 rows):", r[:6].std(1).tolist(), " per-row mean:", r[:6].mean(1).tolist())
print("  rows 32,64,96,127 mean ratio:", [r[i].mean().item() for i in (32, 64, 96, 127)])
xr = x[0].double()
def rstd_of(s, q): 
    mean = s / K
    return 1.0 / torch.sqrt(q / K - mean * mean + 1e-5)
tiles = xr.view(-1, 32)
S, Q = tiles.sum(1), (tiles * tiles).sum(1)
true = rstd_of(S.sum(), Q.sum())
print("observed rstd ratio row0:", r[0].mean().item())
for name, s, q in (("missing last tile", S.sum() - S[-1], Q.sum() - Q[-1]), ("missing tile 0", S.sum() - S[0], Q.sum() - Q[0]),
                   ("tile0 twice, last missing", S.sum() + S[0] - S[-1], Q.sum() + Q[0] - Q[-1]),
                   ("tile0 twice", S.sum() + S[0], Q.sum() + Q[0]), ("tile1 twice", S.sum() + S[1], Q.sum() + Q[1]),
                   ("tile 1 missing", S.sum() - S[1], Q.sum() - Q[1]),
                   ("tile1 twice, tile 0 missing", S.sum() + S[1] - S[0], Q.sum() + Q[1] - Q[0])):
    print(f"  {name:32s} ratio {(rstd_of(s, q) / true).item():.7f}")
# half-tiles (kk): pairs are 16 k wide
h = xr.view(-1, 16); Sh, Qh = h.sum(1), (h * h).sum(1)
for name, s, q in (("first half-tile twice", S.sum() + Sh[0], Q.sum() + Qh[0]), ("second half-tile twice", S.sum() + Sh[1], Q.sum() + Qh[1]),
                   ("first half missing", S.sum() - Sh[0], Q.sum() - Qh[0]), ("second half missing", S.sum() - Sh[1], Q.sum() - Qh[1])):
    print(f"  {name:32s} ratio {(rstd_of(s, q) / true).item():.7f}")
