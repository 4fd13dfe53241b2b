"""GPU, one-rank RCCL communicator: the sharded scoring step as bench.py --gpus N runs it - ONE scorer launch with per-range
arrival counters, a side stream parked on each counter (hipStreamWaitValue32), per-range finish, per-range all_gather on the
collectives' stream - must reproduce the plain one-launch matrix bit for bit, step after step (counters and buffers are reused)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_overlapped_shard_scorer_one_rank_rccl(rccl_comm):
    from dldkd_amd import dist as ddist
    from dldkd_amd import scoring
    nq, nv, L = 5000, 615, 128
    g = torch.Generator(device=DEV).manual_seed(9)
    lens = torch.randint(1, L + 1, (nv,), generator=g, device=DEV)
    mask = (torch.arange(L, device=DEV)[None] < lens[:, None]).float()
    gal = [torch.randn(nv, L, 384, generator=g, device=DEV) * mask[..., None] for _ in range(2)]
    qs = [torch.randn(nq, 384, generator=g, device=DEV) for _ in range(2)]
    pg = scoring.pack_gallery(gal, mask)
    ref = scoring.simpool_eval(scoring.pack_queries(qs), pg)[0]
    try:
        backend = ddist.HipShardBackend(qs, pg, min_ranges=4)
        n = backend.n_ranges
        assert n >= 4 and backend.bounds == ddist.query_ranges(nq, n, backend.per_range)
        ov = ddist.OverlappedShardScorer(backend, backend.bounds, nv, DEV)
        with pytest.raises(ValueError):                         # a split that does not tile the queries is refused up front
            ddist.OverlappedShardScorer(backend, [(0, 100), (200, nq)], nv, DEV)
        for _ in range(3):
            ov.step()
        torch.cuda.synchronize()
        assert torch.equal(ov.assemble(nv), ref)
        assert ov.done.cpu().tolist() == [backend.arrivals] * n
        # new queries through the same object: nothing stale survives a step
        qs[0].copy_(torch.randn(nq, 384, generator=g, device=DEV))
        ov.step()
        torch.cuda.synchronize()
        assert torch.equal(ov.assemble(nv), scoring.simpool_eval(scoring.pack_queries(qs), pg)[0])
    finally:
        torch.cuda.synchronize()
