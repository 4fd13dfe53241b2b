"""K4 in isolation: HBM GB/s of the bf16 input projection on a beyond-Infinity-Cache input (GPU box)."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("dl-dkd_amd", "tests/golden", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
from dldkd_amd import ops
from dldkd_amd.model import DLDKD
DEV = "cuda:0"
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
M = int(sys.argv[2]) if len(sys.argv) > 2 else 400_000
cfg = types.SimpleNamespace(visual_input_size=K, query_input_size=768, inheritance_hidden=384, exploration_hidden=384,
                            max_ctx_l=128, max_desc_l=30, input_drop=0.2, drop=0.2, n_heads=4, initializer_range=0.02,
                            margin=0.1, use_hard_negative=True, hard_pool_size=20, label_style="soft")
opt_ = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                             collection="tvr", alpha=0.8, belta=0.8)
m = DLDKD(cfg, opt_).to(DEV).eval()
x = torch.nn.functional.normalize(torch.randn(M, K, device=DEV), dim=-1)
f = ops.FoldedInProj([m.visual_input_proj, m.exp_visual_input_proj])
byts = M * K * 4 + M * 768 * 4 + 768 * K * 2
flops = 2.0 * M * K * 768
outs = {}
for kern in os.environ.get("K4_KERNELS", "full,rows128").split(","):
    ops.INPROJ_KERNEL = kern
    for _ in range(2): outs[kern] = ops.in_proj_h16(x, f)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
    ev[0].record()
    for i in range(10):
        ops.in_proj_h16(x, f); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(10)); med = ts[5]
    print(f"K4 in_proj_h16[{kern}] M={M} K={K}: {med:.3f} ms  algorithmic {byts/med/1e6:.0f} GB/s ({byts/med/1e6/8000*100:.1f}% of 8 TB/s)  {flops/med/1e9:.0f} TFLOP/s", flush=True)
if len(outs) == 2:
    a, b = outs["full"], outs["rows128"]
    for i in range(2):
        d = (a[i] - b[i]).abs().max().item()
        print(f"   branch {i}: max |full - rows128| = {d:.3e}  (|y| max {a[i].abs().max().item():.3f})  nan {torch.isnan(b[i]).sum().item()}")
ops.INPROJ_KERNEL = "rows128"
with torch.no_grad():
    t = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    xs = x[: 200 * 128].view(200, 128, K)
    t[0].record(); y_ref = m.visual_input_proj(xs); t[1].record()
    torch.cuda.synchronize()
    print(f"   fp32 parity path, same op, one branch, 25600 rows: {t[0].elapsed_time(t[1]):.3f} ms")
