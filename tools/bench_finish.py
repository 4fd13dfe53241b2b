"""Time the finish kernel (un-permute + transpose + 0.7/0.3 fusion) at C2 size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dl-dkd_amd"))
import torch
from dldkd_amd import native
NQ, NV = 10895, 21793
dev = "cuda:0"
L = native.lib()
nqp = (NQ + 31) // 32 * 32
ws = torch.randn(2 * NV * nqp, device=dev)
inv = torch.randperm(NV, device=dev).int()
fused = torch.empty(NQ, NV, device=dev)
def run(): native.check(L.dldkd_simpool_finish(native.ptr(ws), native.ptr(inv), NQ, NV, 2, 0.7, 0.3, native.ptr(fused), None, None, native.stream()), "f")
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"finish {ms:.3f} ms = {(2*NV*nqp*4 + NQ*NV*4)/ms/1e6:.0f} GB/s  checksum {fused.double().sum().item():.6e}")
