"""K1 on ragged galleries, MEASURED: what packing whole videos into a wave's 128 rows could buy, against what
the segment-aware pooling it needs would cost - same box, same process, HIP events around the scorer launch only.

  A  baseline: the C2 gallery (21,793 videos, U{24..128} clips): every video owns a wave and pads its last 16-clip tile
  B  upper bound of scheme (b) of DESIGN 9.5: the videos best-fit-decreasing-packed into 128-row waves, each pack scored as ONE
     video of sum(len) clips - exactly the MFMA tiles, gallery bytes and workgroup count the packed scheme would have, with NO
     second running maximum, no masked boundary tile, no extra store (the scores of a pack are one number, i.e. wrong - this is a
     timing proxy: it bounds the scheme's gain from above)
  P  the scheme as built: PAIR waves (dldkd_simpool_eval_pairs_bf16: two videos per wave, longest with shortest, two running maxima
     routed per 4-row lane group) - real scores, compared bit for bit with A's partial planes
  C  (DIAG build only: make -C dl-dkd_amd/csrc DIAG=1, then DLDKD_SIMPOOL_ABLATE=1) the scorer without its max-pool: what today's
     pooling VALU (16 v_max3 + 4 cross-lane ops per 16-query sub-tile) costs; the packed scheme adds ~29 VALU per sub-tile to those
     ~22 (a wave-uniform switch point, a masked boundary tile, a second cross-lane reduction and store)

    python tools/ablation_simpool_ragged.py [--iters 12] > profiles/r04/ablation_simpool_ragged.json
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dl-dkd_amd"))
from dldkd_amd import scoring  # noqa: E402

NQ, NV, L, LEN_LO = 10895, 21793, 128, 24


def best_fit_decreasing(lens, cap=128):
    """Bins of capacity `cap` rows; every video whole in one bin.  Returns the bins' row sums."""
    import bisect
    free, sums = [], []                       # free: sorted list of (remaining, bin index)
    for n in sorted(lens, reverse=True):
        i = bisect.bisect_left(free, (n, -1))
        if i < len(free):
            rem, b = free.pop(i)
            sums[b] += n
            if rem - n > 0:
                bisect.insort(free, (rem - n, b))
        else:
            sums.append(n)
            if cap - n > 0:
                bisect.insort(free, (cap - n, len(sums) - 1))
    return sums


def time_scorer(lens, nq, iters, dev="cuda:0", seed=2, pairs=False, keep=None):
    gen = torch.Generator(device=dev).manual_seed(seed)
    nv = lens.numel()
    mask = (torch.arange(L, device=dev).unsqueeze(0) < lens.to(dev).unsqueeze(1)).float()
    gs = [torch.randn(nv, L, 384, generator=gen, device=dev) for _ in range(2)]
    pg = scoring.pack_gallery(gs, mask)
    del gs
    pq = scoring.pack_queries([torch.randn(nq, 384, generator=gen, device=dev) for _ in range(2)])
    ws = None
    scoring.PAIR_WAVES = pairs
    for _ in range(3):
        ws = scoring.simpool_partials(pq, pg, ws)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        ws = scoring.simpool_partials(pq, pg, ws)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    scoring.PAIR_WAVES = False
    res = {"ms_median": ts[len(ts) // 2], "ms_min": ts[0], "n_waves": int(nv), "clips": int(lens.sum()),
           "tiles16": int(((lens + 15) // 16).sum())}
    if pairs:
        plan = pg.pair_plan()[0].cpu()
        sl = lens[pg.order.long().cpu()]
        rows = torch.where(plan[:, 1] >= 0, (sl[plan[:, 0].long()] + 3) // 4 * 4 + sl[plan[:, 1].clamp(min=0).long()], sl[plan[:, 0].long()])
        res.update(n_waves=int(plan.shape[0]), tiles16=int(((rows + 15) // 16).sum()), paired_waves=int((plan[:, 1] >= 0).sum()))
    if keep is not None:
        n = 2 * nv * ((nq + 31) // 32 * 32)
        planes = ws.view(torch.float32)[:n]
        if "ref" in keep:
            res["bit_identical_to_A"] = bool(torch.equal(planes.view(torch.int32), keep["ref"].view(torch.int32)))
        else:
            keep["ref"] = planes.clone()
    del pg, pq, ws
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=12)
    a = ap.parse_args()
    g = torch.Generator().manual_seed(2)
    lens = torch.randint(LEN_LO, L + 1, (NV,), generator=g)
    out = {"workload": "C2: 10,895 queries x 21,793 videos x U{24..128} clips x 2 branches, scorer launch only",
           "ablate_env": os.environ.get("DLDKD_SIMPOOL_ABLATE", "0")}
    keep = {}
    out["A_baseline"] = time_scorer(lens, NQ, a.iters, keep=keep)
    out["P_pair_waves"] = time_scorer(lens, NQ, a.iters, pairs=True, keep=keep)
    del keep
    out["A_after_P"] = time_scorer(lens, NQ, a.iters)
    out["P_again"] = time_scorer(lens, NQ, a.iters, pairs=True)
    out["pair_waves_time_saved_pct"] = 100.0 * (1 - (out["P_pair_waves"]["ms_median"] + out["P_again"]["ms_median"]) /
                                                (out["A_baseline"]["ms_median"] + out["A_after_P"]["ms_median"]))
    packs = torch.tensor(best_fit_decreasing(lens.tolist()), dtype=lens.dtype)
    out["B_packed_upper_bound"] = time_scorer(packs, NQ, a.iters)
    out["A_again"] = time_scorer(lens, NQ, a.iters)                     # brackets B: drift of the box between the two
    base = 0.5 * (out["A_baseline"]["ms_median"] + out["A_again"]["ms_median"])
    out["tiles_saved_pct"] = 100.0 * (1 - out["B_packed_upper_bound"]["tiles16"] / out["A_baseline"]["tiles16"])
    out["time_saved_upper_bound_pct"] = 100.0 * (1 - out["B_packed_upper_bound"]["ms_median"] / base)
    print(json.dumps(out, indent=1))
