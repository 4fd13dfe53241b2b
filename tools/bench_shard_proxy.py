"""Single-GPU proxy of the 8-GPU ActivityNet run: one rank's 615-video shard x all 17,505 queries
against the whole 4,917-video gallery on the same GPU; scorer kernel only (HIP events) and scorer + finish."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dl-dkd_amd"))
from dldkd_amd import scoring  # noqa: E402


def time_ms(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return ts[len(ts) // 2]


def shard_proxy(dev="cuda:0", nq=17505, nv=4917, L=128, world=8, seed=4):
    g = torch.Generator(device=dev).manual_seed(seed)
    gal = [torch.randn(nv, L, 384, generator=g, device=dev) for _ in range(2)]
    qs = [torch.randn(nq, 384, generator=g, device=dev) for _ in range(2)]
    pq = scoring.pack_queries(qs)
    pg_full = scoring.pack_gallery(gal, None)
    s = (nv + world - 1) // world
    pg_sh = scoring.pack_gallery([x[:s] for x in gal], None)
    del gal
    out = {"config": f"{nv} videos x {L} clips (all valid) x {nq} queries, 2 branches; shard = {s} videos (1 of {world})"}
    ws = {}

    def k(pg, split, key):
        def f():
            ws[key] = scoring.simpool_partials(pq, pg, ws.get(key), q_split=split)
        return f

    def kf(pg, split, key):
        def f():
            ws[key] = scoring.simpool_partials(pq, pg, ws.get(key), q_split=split)
            scoring.simpool_finish(ws[key], pq, pg)
        return f
    out["full_planned_split"] = scoring.plan_query_split(nq, nv, 2)[0]
    out["shard_planned_split"] = scoring.plan_query_split(nq, s, 2)[0]
    out["full_kernel_ms_split1"] = time_ms(k(pg_full, 1, "f"))
    out["full_kernel_ms"] = time_ms(k(pg_full, 0, "f"))
    out["shard_kernel_ms_split1"] = time_ms(k(pg_sh, 1, "s"))
    out["shard_kernel_ms"] = time_ms(k(pg_sh, 0, "s"))
    for sp in (2, 3, 4, 5, 6, 8, 12, 16):
        out[f"shard_kernel_ms_split{sp}"] = time_ms(k(pg_sh, sp, "s"))
    out["full_step_ms"] = time_ms(kf(pg_full, 0, "f"))
    out["shard_step_ms"] = time_ms(kf(pg_sh, 0, "s"))
    out["shard_over_full_kernel"] = out["shard_kernel_ms"] / out["full_kernel_ms"]
    out["shard_over_full_kernel_round1_grid"] = out["shard_kernel_ms_split1"] / out["full_kernel_ms_split1"]
    out["shard_over_full_step"] = out["shard_step_ms"] / out["full_step_ms"]
    return out


if __name__ == "__main__":
    print(json.dumps(shard_proxy(), indent=1))
