"""GPU, BASELINE.json configs[3] size (ActivityNet: 4,917 videos x 128 clips ALL valid x 17,505 queries x 2 branches,
/root/reference/do_activitynet.sh:6-12; the gallery loop being sharded is method/eval.py:188-212): the size-independent
properties of tests/test_fullsize_properties_gpu.py at C4, plus what C4 adds - the 8-way gallery sharding with the
query-split grid (one rank's shard = 615 videos = 308 workgroups on 256 CUs) and the per-range finish that the
overlapped all-gather uses.

  P1 video-order invariance      P2 8-shard assembly (each shard scored with its planned query split), bit for bit
  P3 query-split invariance      P4 truncation monotonicity      P5 fusion identity
  P6 planted ground truth        P7 sampled oracle parity (fp32 oracle, bf16 tolerance)
  P8 per-range finish + arrival counters of one sharded launch = the one-launch matrix, bit for bit
"""
import numpy as np
import pytest
import torch

import dldkd_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NQ, NV, L, D = 17505, 4917, 128, 384


@pytest.fixture(scope="module")
def c4():
    from dldkd_amd import scoring
    g = torch.Generator(device=DEV).manual_seed(4)
    gs = [torch.randn(NV, L, D, generator=g, device=DEV) for _ in range(2)]
    gt = torch.arange(NQ, device=DEV) % NV
    clip = torch.randint(0, L, (NQ,), generator=g, device=DEV)
    # planted queries (sigma = 0): exact copies of one clip of the GT video, per branch
    qs = [x[gt, clip].clone() for x in gs]
    pg = scoring.pack_gallery(gs, None)
    pq = scoring.pack_queries(qs)
    fused, s0, s1 = scoring.simpool_eval(pq, pg, want_branches=True)
    return dict(gs=gs, qs=qs, gt=gt, pg=pg, pq=pq, fused=fused, s0=s0, s1=s1)


def test_p1_video_order_invariance(c4):
    from dldkd_amd import scoring
    perm = torch.randperm(NV, generator=torch.Generator().manual_seed(1)).to(DEV)
    pg = scoring.pack_gallery([x[perm] for x in c4["gs"]], None)
    fused, _, _ = scoring.simpool_eval(c4["pq"], pg)
    assert torch.equal(fused, c4["fused"][:, perm])


def test_p2_eight_shard_assembly_with_planned_split(c4):
    from dldkd_amd import scoring
    shard = (NV + 7) // 8
    assert shard == 615
    cols = []
    for r in range(8):
        lo, hi = r * shard, min((r + 1) * shard, NV)
        pg = scoring.pack_gallery([x[lo:hi] for x in c4["gs"]], None)
        n, per = scoring.plan_query_split(NQ, hi - lo, 2)
        assert n >= 3                                  # 308 workgroups per range: the shard needs the split
        ws = scoring.simpool_partials(c4["pq"], pg, q_split=n)
        cols.append(scoring.simpool_finish(ws, c4["pq"], pg)[0])
    assert torch.equal(torch.cat(cols, 1), c4["fused"])


def test_p3_query_split_invariance(c4):
    from dldkd_amd import scoring
    for split in (1, 2, 7, 16):
        ws = scoring.simpool_partials(c4["pq"], c4["pg"], q_split=split)
        assert torch.equal(scoring.simpool_finish(ws, c4["pq"], c4["pg"])[0], c4["fused"]), split


def test_p4_truncation_monotonicity(c4):
    from dldkd_amd import scoring
    lens = torch.randint(1, L + 1, (NV,), generator=torch.Generator().manual_seed(3)).to(DEV)
    mask = (torch.arange(L, device=DEV)[None] < lens[:, None]).float()
    pg = scoring.pack_gallery([x * mask[..., None] for x in c4["gs"]], mask)
    _, t0, t1 = scoring.simpool_eval(c4["pq"], pg, want_branches=True)
    assert bool((t0 <= c4["s0"]).all()) and bool((t1 <= c4["s1"]).all())
    assert bool((t0 < c4["s0"]).any())


def test_p5_fusion_identity(c4):
    ref = 0.7 * c4["s0"].double() + 0.3 * c4["s1"].double()
    assert (c4["fused"].double() - ref).abs().max().item() <= 1.2e-7


def test_p6_planted_ground_truth_ranks_first(c4):
    from dldkd_amd import eval as ev
    gts = c4["fused"][torch.arange(NQ, device=DEV), c4["gt"]]
    assert (gts - 1.0).abs().max().item() < 2e-3
    t2v = {q: [int(v)] for q, v in enumerate(c4["gt"].cpu().tolist())}
    r1, r5, r10, r100, medr, meanr = ev.eval_q2m(-c4["fused"], t2v)
    assert r1 == 100.0 and medr == 1.0


def test_p7_sampled_oracle_parity(c4):
    rs = np.random.RandomState(7)
    qi = torch.from_numpy(rs.choice(NQ, 20, replace=False)).to(DEV)
    vi = torch.from_numpy(rs.choice(NV, 300, replace=False)).to(DEV)
    m = torch.ones(300, L)
    oi, oe = orc.eval_scores(c4["qs"][0][qi].cpu(), c4["qs"][1][qi].cpu(), c4["gs"][0][vi].cpu(), c4["gs"][1][vi].cpu(), m)
    ref = orc.fuse_scores(oi, oe)
    got = c4["fused"][qi][:, vi].cpu()
    assert (got - ref).abs().max().item() < 6e-3


def test_p8_per_range_finish_of_one_sharded_launch(c4):
    """What OverlappedShardScorer does on every rank: ONE launch over all queries with arrival counters, then one finish
    per query range (here on the same stream; the stream-ordered wait is exercised in tests/test_dist_gpu.py)."""
    from dldkd_amd import scoring
    lo, hi = 3 * 615, 4 * 615
    pg = scoring.pack_gallery([x[lo:hi] for x in c4["gs"]], None)
    n, per = scoring.plan_query_split(NQ, hi - lo, 2, min_split=4)
    done = torch.zeros(n, dtype=torch.int32, device=DEV)
    ws = scoring.simpool_partials(c4["pq"], pg, q_split=n, done=done)
    rows = [scoring.simpool_finish(ws, c4["pq"], pg, q_range=(a, min(a + per, NQ)))[0] for a in range(0, NQ, per)]
    assert len(rows) == n
    assert torch.equal(torch.cat(rows, 0), c4["fused"][:, lo:hi])
    assert done.cpu().tolist() == [(615 + 3) // 4 * 2] * n
