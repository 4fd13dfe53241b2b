"""Diagnostic: per-workgroup timeline of in_proj_rows128_kernel from dldkd_debug_in_proj_rows128_timeline (s_memtime /
s_memrealtime stamps at kernel start, k-loop start, k-loop end, kernel end; the in-kernel clock under load)."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dl-dkd_amd"))
import torch, numpy as np
from dldkd_amd import ops
from dldkd_amd.model import DLDKD
K, M = 3072, 400000
cfg = types.SimpleNamespace(visual_input_size=K, query_input_size=768, inheritance_hidden=384, exploration_hidden=384,
                            max_ctx_l=128, max_desc_l=30, input_drop=0.2, drop=0.2, n_heads=4, initializer_range=0.02,
                            margin=0.1, use_hard_negative=True, hard_pool_size=20, label_style="soft")
opt_ = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                             collection="tvr", alpha=0.8, belta=0.8)
m = DLDKD(cfg, opt_).to("cuda:0").eval()
x = torch.nn.functional.normalize(torch.randn(M, K, device="cuda:0"), dim=-1)
f = ops.FoldedInProj([m.visual_input_proj, m.exp_visual_input_proj])
from dldkd_amd import native
L = native.lib()
fo = f.get()
ntile = (M + 127) // 128
stamps = torch.zeros(ntile, 12, dtype=torch.int64, device="cuda:0")      # one row per (persistent) workgroup; unused rows stay 0
ys = [torch.empty(M, 384, device="cuda:0") for _ in range(2)]
for _ in range(20):
    native.check(L.dldkd_debug_in_proj_rows128_timeline(native.ptr(x), native.ptr(fo.Wf), native.ptr(fo.cs), native.ptr(fo.bb),
                                                        native.ptr(ys[0]), native.ptr(ys[1]), M, K, 1e-5, 1, native.ptr(stamps),
                                                        native.stream()), "timeline")
torch.cuda.synchronize()
raw = stamps.cpu().numpy()
raw = raw[raw[:, 1] != 0]
t0 = raw[:, 1].min()
rt = (raw[:, [1, 3, 5, 7, 9]] - t0) / 100.0      # us (100 MHz): start, first loop start, first loop end, first epilogue end, end
cyc = raw[:, [2, 4, 6, 8]] - raw[:, [0, 2, 4, 6]]
dur = rt[:, 1:] - rt[:, :-1]
ntl = raw[:, 11]
print("workgroups", len(raw), "tiles per workgroup", int(ntl.min()), "-", int(ntl.max()), " kernel span us", rt[:, 4].max())
print("clock GHz (first loop): median", np.median(cyc[:, 1] / dur[:, 1]) / 1e3)
for name, i in (("prologue", 0), ("first loop", 1), ("first epilogue", 2)):
    print(f"{name:15s} us: median {np.median(dur[:, i]):8.2f}  p10 {np.percentile(dur[:, i], 10):8.2f}  p90 {np.percentile(dur[:, i], 90):8.2f}   cycles median {np.median(cyc[:, i]):10.0f}")
rest = (rt[:, 4] - rt[:, 3]) / np.maximum(ntl - 1, 1)
print(f"later tiles     us per tile (loop + boundary): median {np.median(rest):8.2f}  p10 {np.percentile(rest, 10):8.2f}  p90 {np.percentile(rest, 90):8.2f}"
      f"   cycles median {np.median(cyc[:, 3] / np.maximum(ntl - 1, 1)):10.0f}")
print("workgroups per XCC", np.bincount((raw[:, 10] & 0xf).astype(int)))
