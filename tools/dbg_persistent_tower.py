import os, sys
import numpy as np, torch
sys.path[:0] = ["/root/repo/dl-dkd_amd", "/root/repo/tests", "/root/repo/tests/golden", "/root/repo/oracle"]
import test_tower_seq_gpu as Tt
from dldkd_amd import ops, scoring
DEV = "cuda:0"; H = 384
g = torch.Generator().manual_seed(77)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1700
lens = torch.randint(1, 129, (n,), generator=g).tolist()
ts, packs, _, lens_t = Tt._setup(seed=4, lens=lens)
rows = int(sum(lens))
h16 = [torch.relu(torch.randn(rows, H, generator=g)).half().to(DEV) for _ in range(2)]
row0 = torch.tensor([0] + np.cumsum(lens)[:-1].tolist(), dtype=torch.int32, device=DEV)
items_np = ops.plan_tower_items(lens_t.numpy())
items = torch.from_numpy(items_np).to(DEV)
out = []
for hs in (h16, [x.float() for x in h16]):
    pk = scoring.GalleryPacker(n, 128, 2, torch.device(DEV))
    for blob in pk.blobs: blob.fill_(0x55)
    pk.reserve(n, 128)
    ops.tower_seq(hs, packs, lens_t.to(DEV), seq_rows=0, row0=row0, items=items, out_mode=1, gallery=pk.blobs, v0=0, Lp=pk.Lp, lens_out=pk.lens)
    torch.cuda.synchronize()
    out.append([b.clone() for b in pk.blobs])
item_of = {}
for it in range(items_np.shape[0]):
    for w in range(4):
        e = int(items_np[it, w])
        if e >= 0: item_of.setdefault(e >> 10, it)
for b in range(2):
    a, c = out[0][b].view(n, -1), out[1][b].view(n, -1)
    bad = (a != c).any(1).cpu().numpy()
    good_items = sorted({item_of[v] for v in range(n) if not bad[v]})
    bad_items = sorted({item_of[v] for v in range(n) if bad[v]})
    print("branch", b, "items", items_np.shape[0], "bad videos", int(bad.sum()), "good items (first 40)", good_items[:40], "...", good_items[-10:])
    untouched = (a == 0x55).all(1).cpu().numpy()
    print("   videos never written:", int(untouched.sum()), " bad items first 20", bad_items[:20])
    # how different: fraction of differing bytes in a bad video
    v = int(np.nonzero(bad)[0][0])
    print("   first bad video", v, "item", item_of[v], "len", lens[v], "differing bytes", int((a[v] != c[v]).sum()), "of", a.shape[1])
for b in range(2):
    a, c = out[0][b].view(n, -1), out[1][b].view(n, -1)
    bad = (a != c).any(1).cpu().numpy()
    for v in np.nonzero(bad)[0][:12]:
        it = item_of[int(v)]
        ents = [int(x) for x in items_np[it]]
        dec = [(e >> 10, (e >> 8) & 3, e & 255) if e >= 0 else None for e in ents]
        d = (a[v] != c[v]).view(128, -1).any(1).nonzero().flatten().tolist()
        print("branch", b, "video", int(v), "len", lens[int(v)], "item", it, "iter", it // 128, "slots", dec, "bad rows", d[:6], "..", d[-3:], len(d))
for b in range(2):
    a = out[0][b].view(torch.bfloat16).view(n, 128, H).float(); c = out[1][b].view(torch.bfloat16).view(n, 128, H).float()
    bad = (a != c).any(2).nonzero()
    for v, rrow in bad[:8].tolist():
        d = (a[v, rrow] - c[v, rrow])
        print("  b", b, "v", v, "row", rrow, "n diff", int((d != 0).sum()), "max", float(d.abs().max()), "norms", float(a[v, rrow].norm()), float(c[v, rrow].norm()), "first idx", (d != 0).nonzero().flatten()[:8].tolist())
