"""Time the GEMM shapes of the C3 training step / eval towers in fp32 and bf16 mode (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dl-dkd_amd"))
import torch
from dldkd_amd import ops
DEV = "cuda:0"
# (label, M, N, K, a_kmajor, b_kmajor): C[M,N] = sum_k A(m,k) B(n,k)
SHAPES = [("fwd 16384x384x384", 16384, 384, 384, 0, 0), ("fwd qkv 16384x1152x384", 16384, 1152, 384, 0, 0),
          ("fwd inproj 16384x384x3072", 16384, 384, 3072, 0, 0), ("dX 16384x384x384", 16384, 384, 384, 0, 1),
          ("dX inproj 16384x3072x384", 16384, 3072, 384, 0, 1), ("dW 384x384x16384", 384, 384, 16384, 1, 1),
          ("dW qkv 1152x384x16384", 1152, 384, 16384, 1, 1), ("dW inproj 384x3072x16384", 384, 3072, 16384, 1, 1),
          ("clip 640x16384x384", 640, 16384, 384, 0, 0), ("dQ 640x384x16384", 640, 384, 16384, 0, 1),
          ("dC 16384x384x640", 16384, 384, 640, 1, 1)]
def run(prec):
    ops.set_gemm_precision(prec)
    for lab, M, N, K, ak, bk in SHAPES:
        a = torch.randn((K, M) if ak else (M, K), device=DEV)
        b = torch.randn((K, N) if bk else (N, K), device=DEV)
        f = (lambda: ops.gemm(a, b, bool(ak), bool(bk), M, N, K)) if (ak or bk) else (lambda: ops.linear(a, b))
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{prec} {lab:28s} {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TF  {(M*K+N*K+M*N)*4/us/1e3:7.1f} GB/s min-traffic")
for prec in sys.argv[1:] or ["fp32", "fp32x3", "bf16"]:
    run(prec)
