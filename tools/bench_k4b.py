"""K4b (in_proj_rows128b: resident bf16 rows + row statistics) against K4 (fp32 rows): agreement on ragged tables, then the
timing of both at the bench shape (400,000 rows x 3072 -> 2 x 384).  `python tools/bench_k4b.py 3072 pmc` runs K4b alone."""
import sys, torch, time
sys.path.insert(0, "/root/repo/dl-dkd_amd")
from dldkd_amd import ops, model_components as mc
dev = "cuda:0"
torch.manual_seed(0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
layers = [mc.LinearLayer(K, 384, layer_norm=True, dropout=0.0, relu=True).to(dev) for _ in range(2)]
for l in layers:
    torch.nn.init.normal_(l.LayerNorm.weight, 1.0, 0.2); torch.nn.init.normal_(l.LayerNorm.bias, 0.0, 0.2)
fold = ops.FoldedInProj(layers)
mode = sys.argv[2] if len(sys.argv) > 2 else "all"
for (n, L) in ((5, 32), (37, 128), (300, 96)) if mode == "all" else ():
    lens = torch.randint(1, L + 1, (n,)).numpy()
    x = torch.randn(n, L, K, device=dev) * (1 + torch.rand(n, L, 1, device=dev)) + 0.3
    tab = ops.ResidentRows(K, dev)
    tab.append(x[: n // 2], lens[: n // 2]); tab.append(x[n // 2:], lens[n // 2:])
    rows = torch.cat([x[i, :lens[i]] for i in range(n)], 0)
    assert tab.rows == rows.shape[0]
    assert torch.equal(tab.xb[:tab.rows], rows.to(torch.float16))
    mu = rows.double().mean(1); var = rows.double().var(1, unbiased=False)
    print("stats err", (tab.mean[:tab.rows].double() - mu).abs().max().item(), (tab.rstd[:tab.rows].double() * (var + 1e-5).sqrt() - 1).abs().max().item())
    with torch.no_grad():
        y_ref = ops.in_proj_h16(rows.contiguous(), fold)
        y = ops.in_proj_resident(tab, 0, tab.rows, fold)
        torch.cuda.synchronize()
        for b in range(2):
            d = (y[b] - y_ref[b]).abs().max().item()
            # fp64 reference with the kernel's roundings: bf16 x, bf16 W', fp32-ish stats
            l = layers[b]
            Wp = (l.net[1].weight * l.LayerNorm.weight).to(torch.float16).double()
            xr = rows.to(torch.float16).double()
            ref = (xr @ Wp.T - mu[:, None] * Wp.sum(1)[None]) / (var + 1e-5).sqrt()[:, None] + (l.net[1].weight.double() @ l.LayerNorm.bias.double() + l.net[1].bias.double())
            ref = ref.clamp_min(0)
            print(n, L, "branch", b, "max |K4b - K4|", d, "max |K4b - fp64|", (y[b].double() - ref).abs().max().item(), "scale", ref.abs().max().item())
        # a sub-range in the middle (not tile aligned)
        lo, hi = 7, tab.rows - 3
        ys = ops.in_proj_resident(tab, lo, hi, fold)
        print("subrange equal:", all(torch.equal(ys[b], y[b][lo:hi]) for b in range(2)))
# timing at the bench shape
M = 400000
x = torch.randn(3125, 128, K, device=dev)
tab = ops.ResidentRows(K, dev, M)
tab.append(x, [128] * 3125)
x2 = x.reshape(-1, K)
outs = [torch.empty(M, 384, device=dev) for _ in range(2)]
def tm(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        f(); ev[i + 1].record()
    torch.cuda.synchronize()
    t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return t[n // 2]
if mode == "pmc":
    with torch.no_grad():
        for _ in range(5): ops.in_proj_resident(tab, 0, M, fold, out=outs)
    torch.cuda.synchronize()
    sys.exit(0)
with torch.no_grad():
    t_b = tm(lambda: ops.in_proj_resident(tab, 0, M, fold, out=outs))
    t_a = tm(lambda: ops.in_proj_h16(x2, fold))
fl = 2.0 * M * K * 768
print(f"K4 {t_a:.3f} ms ({fl/t_a/1e9:.0f} TFLOP/s)  K4b {t_b:.3f} ms ({fl/t_b/1e9:.0f} TFLOP/s)")
