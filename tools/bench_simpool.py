"""Quick HIP-event timing of the scorer at a given shape (GPU box).  Not the contract bench (bench.py)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dl-dkd_amd"))
from dldkd_amd import scoring  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nq", type=int, default=10895)
ap.add_argument("--nv", type=int, default=21793)
ap.add_argument("--L", type=int, default=128)
ap.add_argument("--len-lo", type=int, default=24)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--split", type=int, default=0, help="query ranges of the launch grid (0 = planned)")
a = ap.parse_args()
dev = "cuda:0"
gen = torch.Generator(device=dev).manual_seed(2)
lens = torch.randint(a.len_lo, a.L + 1, (a.nv,), generator=gen, device=dev)
mask = (torch.arange(a.L, device=dev).unsqueeze(0) < lens.unsqueeze(1)).float()
blobs_g, blobs_q = [], []
t0 = time.time()
gs = []
for b in range(2):
    g = torch.randn(a.nv, a.L, 384, generator=gen, device=dev)
    gs.append(g)
pg = scoring.pack_gallery(gs, mask)
del gs, g
qs = [torch.randn(a.nq, 384, generator=gen, device=dev) for _ in range(2)]
pq = scoring.pack_queries(qs)
torch.cuda.synchronize()
print(f"setup {time.time()-t0:.1f}s  sum(len)={int(lens.sum())}")
def run():
    ws = scoring.simpool_partials(pq, pg, run.ws, q_split=a.split)
    run.ws = ws
    return scoring.simpool_finish(ws, pq, pg)


run.ws = None
print("planned split", scoring.plan_query_split(a.nq, a.nv, 2), "used", a.split or "planned")
for _ in range(2):
    run()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
ev[0].record()
for i in range(a.iters):
    run()
    ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.iters))
med = ts[len(ts) // 2]
flops = 2.0 * 384 * 2 * a.nq * float(lens.sum())
print(f"ms/iter median {med:.3f} min {ts[0]:.3f} max {ts[-1]:.3f}  pairs/s {a.nq*a.nv/med*1e3:.3e}  "
      f"algorithmic TFLOP/s {flops/med/1e9:.1f}  ({flops/med/1e9/2500*100:.1f}% of 2.5 PF)")
