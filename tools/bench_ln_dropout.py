"""layernorm_kernel on 3072-wide rows (the training input projection's LayerNorm + dropout, bf16 rows out): median us of 20 launches."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dl-dkd_amd"))
import torch

from dldkd_amd import native

L = native.lib()
dev = "cuda:0"
M, K = 16384, 3072
x = torch.randn(M, K, device=dev)
g, b = torch.ones(K, device=dev), torch.zeros(K, device=dev)
z16 = torch.empty(M, K, dtype=torch.bfloat16, device=dev)
z32 = torch.empty(M, K, device=dev)
keep = torch.empty(M, K, dtype=torch.uint8, device=dev)
stats = torch.empty(2, M, device=dev)
lens = torch.randint(24, 129, (M // 128,), device=dev)
mask = (torch.arange(128, device=dev)[None] < lens[:, None]).float().reshape(-1).contiguous()
gfl = torch.empty(M // 32, dtype=torch.uint8, device=dev)
p = native.ptr


def tm(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        f()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))[n // 2] * 1e3


s = native.stream()
call = lambda kp, pd, rm, gf: L.dldkd_layernorm_dropout_bf16(p(x), p(g), p(b), p(z16), kp, p(stats), M, K, 1e-5, pd, 1, 0, None, rm, gf, s)   # noqa: E731
print("bf16 rows, p=0.2, keep bytes       : %.1f us" % tm(lambda: call(p(keep), 0.2, None, None)))
print("bf16 rows, p=0.2, no keep bytes    : %.1f us" % tm(lambda: call(None, 0.2, None, None)))
print("bf16 rows, p=0.2, no keep, U{24..128} of 128 rows valid: %.1f us" % tm(lambda: call(None, 0.2, p(mask), p(gfl))))
print("bf16 rows, p=0                     : %.1f us" % tm(lambda: call(None, 0.0, None, None)))
print("copy 201 MB                        : %.1f us" % tm(lambda: z32.copy_(x)))
