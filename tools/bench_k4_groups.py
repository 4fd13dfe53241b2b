"""K4 (in_proj_rows128_kernel) dense vs row-group table: HIP-event medians, interleaved, one process."""
import os, sys, types, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dl-dkd_amd"))
import numpy as np, torch
from dldkd_amd import ops
DEV = "cuda:0"
K = 3072
torch.manual_seed(0)
layers = [torch.nn.Module() for _ in range(2)]
for l in layers:
    l.LayerNorm = torch.nn.LayerNorm(K).to(DEV)
    l.net = torch.nn.Sequential(torch.nn.Dropout(0.0), torch.nn.Linear(K, 384).to(DEV))
fold = ops.FoldedInProj(layers)
n, L = 1024, 128
x = torch.nn.functional.normalize(torch.randn(n, L, K, device=DEV), dim=-1)
g = torch.Generator().manual_seed(1)
lens = torch.randint(24, 129, (n,), generator=g).numpy()
variants = {"dense_1024_tiles": dict(x=x, groups=None), "dense_768_tiles": dict(x=x[:768], groups=None), "dense_512_tiles": dict(x=x[:512], groups=None),
            "groups_ragged_U24_128": dict(x=x, groups=torch.from_numpy(ops.plan_row_groups(lens, L)).to(DEV)),
            "groups_all_rows": dict(x=x, groups=torch.from_numpy(ops.plan_row_groups(np.full(n, 128), L)).to(DEV)),
            "groups_first_768_videos": dict(x=x, groups=torch.from_numpy(ops.plan_row_groups(np.where(np.arange(n) < 768, 128, 0), L)).to(DEV))}
times = {k: [] for k in variants}
for k, v in variants.items():
    ops.in_proj_h16(v["x"], fold, groups=v["groups"])
torch.cuda.synchronize()
for _ in range(15):
    for k, v in variants.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.in_proj_h16(v["x"], fold, groups=v["groups"]); e1.record(); e1.synchronize()
        times[k].append(e0.elapsed_time(e1))
out = {k: {"ms_median": float(np.median(t)), "ms_min": float(min(t)), "tiles": int(v["groups"].numel() // 4 if v["groups"] is not None else v["x"].shape[0])} for (k, t), v in zip(times.items(), variants.values())}
print(json.dumps(out))
