#!/bin/bash
# K1 pair waves: parity on small ragged cases + the C2-ragged A/B
cd /root/repo
python - <<'PY' > gpurun_out/r04_k1p_parity.log 2>&1
import sys, torch
sys.path.insert(0, "dl-dkd_amd")
from dldkd_amd import scoring
dev = "cuda:0"
ok = True
for (nq, nv, L, lo, seed) in [(100, 37, 128, 0, 0), (257, 301, 128, 1, 1), (64, 5, 32, 3, 2), (1000, 2000, 128, 24, 3), (33, 1, 128, 128, 4),
                              (500, 777, 100, 1, 5), (129, 64, 64, 60, 6)]:
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(lo, L + 1, (nv,), generator=g)
    mask = (torch.arange(L)[None, :] < lens[:, None]).float().to(dev)
    gs = [torch.randn(nv, L, 384, generator=g).to(dev) for _ in range(2)]
    qs = [torch.randn(nq, 384, generator=g).to(dev) for _ in range(2)]
    pg, pq = scoring.pack_gallery(gs, mask), scoring.pack_queries(qs)
    scoring.PAIR_WAVES = False
    a = scoring.simpool_partials(pq, pg).clone()
    scoring.PAIR_WAVES = True
    b = scoring.simpool_partials(pq, pg)
    scoring.PAIR_WAVES = False
    n = 2 * nv * ((nq + 31) // 32 * 32)
    same = torch.equal(a.view(torch.int32)[:n], b.view(torch.int32)[:n])
    plan = pg.pair_plan()
    print(f"nq {nq} nv {nv} L {L} lens>={lo}: waves {plan[1]} (paired {(plan[0][:,1] >= 0).sum().item()})  bit-identical {same}")
    ok &= same
print("ALL OK" if ok else "MISMATCH")
PY
tail -12 gpurun_out/r04_k1p_parity.log
python tools/ablation_simpool_ragged.py --iters 12 > gpurun_out/r04_k1p_ab.json 2> gpurun_out/r04_k1p_ab.err
tail -5 gpurun_out/r04_k1p_ab.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_k1p_ab.json"))
for k, v in d.items():
    print(k, v)
PY
