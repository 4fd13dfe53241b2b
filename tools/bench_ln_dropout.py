import sys, torch
sys.path.insert(0, "/root/repo/dl-dkd_amd")
from dldkd_amd import native
L = native.lib()
dev = "cuda:0"
M, K = 16384, 3072
x = torch.randn(M, K, device=dev); g = torch.ones(K, device=dev); b = torch.zeros(K, device=dev)
z16 = torch.empty(M, K, dtype=torch.bfloat16, device=dev); z32 = torch.empty(M, K, device=dev)
keep = torch.empty(M, K, dtype=torch.uint8, device=dev); stats = torch.empty(2, M, device=dev)
def tm(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        f(); ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))[n // 2] * 1e3
s = native.stream()
print("bf16 out, p=0.2: %.1f us" % tm(lambda: L.dldkd_layernorm_dropout_bf16(native.ptr(x), native.ptr(g), native.ptr(b), native.ptr(z16), native.ptr(keep), native.ptr(stats), M, K, 1e-5, 0.2, 1, 0, None, s)))
print("bf16 out, p=0  : %.1f us" % tm(lambda: L.dldkd_layernorm_dropout_bf16(native.ptr(x), native.ptr(g), native.ptr(b), native.ptr(z16), None, native.ptr(stats), M, K, 1e-5, 0.0, 1, 0, None, s)))
print("fp32 out, p=0.2: %.1f us" % tm(lambda: L.dldkd_layernorm_dropout_f32(native.ptr(x), None, 0, native.ptr(g), native.ptr(b), native.ptr(z32), native.ptr(keep), M, K, 1e-5, 0.2, 1, 0, None, s)))
print("fp32 out, p=0  : %.1f us" % tm(lambda: L.dldkd_layernorm_f32(native.ptr(x), None, 0, native.ptr(g), native.ptr(b), native.ptr(z32), M, K, 1e-5, s)))
print("copy 200MB     : %.1f us" % tm(lambda: z32.copy_(x)))
