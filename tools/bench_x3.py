"""Parity-grade fused input projection (in_proj_rows128x3_kernel) vs the LayerNorm + gemm_f32x3 path and fp64 math; with M >= 100000 also timed.
    python tools/bench_x3.py [K] [M]"""
import os, sys, types, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("dl-dkd_amd", "tests/golden", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
from dldkd_amd import ops
from dldkd_amd.model import DLDKD
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
cfg = types.SimpleNamespace(visual_input_size=K, query_input_size=768, inheritance_hidden=384, exploration_hidden=384,
                            max_ctx_l=128, max_desc_l=30, input_drop=0.2, drop=0.2, n_heads=4, initializer_range=0.02,
                            margin=0.1, use_hard_negative=True, hard_pool_size=20, label_style="soft")
opt_ = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                             collection="tvr", alpha=0.8, belta=0.8)
torch.manual_seed(0)
m = DLDKD(cfg, opt_).to("cuda:0").eval()
with torch.no_grad():
    for l in (m.visual_input_proj, m.exp_visual_input_proj):
        l.LayerNorm.weight.add_(0.1 * torch.randn_like(l.LayerNorm.weight)); l.LayerNorm.bias.add_(0.1 * torch.randn_like(l.LayerNorm.bias))
g = torch.Generator(device="cuda:0").manual_seed(1)
x = torch.nn.functional.normalize(torch.randn(M, K, device="cuda:0", generator=g).abs() + 0.1 * torch.randn(M, K, device="cuda:0", generator=g), dim=-1)
f = ops.FoldedInProjX3([m.visual_input_proj, m.exp_visual_input_proj])
with torch.no_grad():
    ys = ops.in_proj_x3(x, f)
    torch.cuda.synchronize()
    for br, l in enumerate((m.visual_input_proj, m.exp_visual_input_proj)):
        ref = l(x)                                   # the existing parity path (LayerNorm kernel + gemm_f32x3)
        xd = x.double()
        xn = torch.nn.functional.layer_norm(xd, (K,), l.LayerNorm.weight.double(), l.LayerNorm.bias.double(), 1e-5)
        ref64 = torch.relu(xn @ l.net[1].weight.double().t() + l.net[1].bias.double())
        d1 = (ys[br] - ref).abs().max().item(); d2 = (ys[br].double() - ref64).abs().max().item(); d3 = (ref.double() - ref64).abs().max().item()
        print(f"branch {br}: |new - parity path| {d1:.3e}   |new - fp64| {d2:.3e}   |parity path - fp64| {d3:.3e}   |y| max {ref64.abs().max().item():.3f}  nan {torch.isnan(ys[br]).sum().item()}")
    if M >= 100000:
        for name, fn in (("x3 fused (stats + kernel)", lambda: ops.in_proj_x3(x, f)), ("parity path (LN + 2 x gemm_f32x3)", lambda: [l(x) for l in (m.visual_input_proj, m.exp_visual_input_proj)])):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            print(f"  {name}: {e0.elapsed_time(e1) / 5:.3f} ms")
