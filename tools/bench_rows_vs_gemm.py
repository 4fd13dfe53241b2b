import sys, torch, time
sys.path.insert(0, "/root/repo/dl-dkd_amd")
from dldkd_amd import ops
dev = "cuda:0"
ops.set_gemm_precision("bf16")
def tm(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        f(); ev[i + 1].record()
    torch.cuda.synchronize()
    t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return t[n // 2] * 1e3
with torch.no_grad():
    for M in (16384, 19200):
        for K, nl in ((384, 1), (384, 3), (768, 1), (3072, 1), (1152, 1)):
            lins = [torch.nn.Linear(K, 384).to(dev) for _ in range(nl)]
            pk = ops.PackedLinear(lins)
            x = torch.randn(M, K, device=dev)
            w = torch.cat([l.weight for l in lins], 0).contiguous(); b = torch.cat([l.bias for l in lins], 0).contiguous()
            t_rows = tm(lambda: ops.linear_rows(x, pk))
            t_gemm = tm(lambda: ops.linear(x, w, b))
            print(f"M={M} K={K} N={384*nl}: rows {t_rows:.1f} us, gemm_bf16 {t_gemm:.1f} us")
