"""CPU, gloo, world_size 2: the N>1 plumbing of the path (sharding, score all-gather assembly, gather-free
ranking, flat gradient bucket).  The scorer/count are injected (the oracle), since there is no GPU here; the
collectives, shard arithmetic and padding are the code under test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import dldkd_oracle as orc
import synth


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, nv, nq, ret):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in ("dl-dkd_amd", "oracle", "tests/golden"):
        sys.path.insert(0, os.path.join(root, p))
    from dldkd_amd import dist as ddist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    d = synth.make_gallery(77, nq, nv, 16, 3, sigma=3.0)
    lo, hi, s = ddist.shard_range(nv, rank, world)
    g, mask = d["g"][lo:hi], d["mask"][lo:hi]
    if hi - lo < s:                                   # pad the short shard like bench.py does
        pad = s - (hi - lo)
        g = torch.cat([g, torch.zeros(pad, 16, 384)]); mask = torch.cat([mask, torch.zeros(pad, 16)]); mask[hi - lo:, 0] = 1
    local = orc.sim_scores(d["q"], g, mask)[0]        # (Nq, S): the injected scorer
    full = ddist.gather_scores(local, nv)
    ranks = ddist.sharded_gt_ranks(local, d["gt"], nv, count_fn=lambda sc, thr, n: (sc[:, :n] > thr[:, None]).sum(1).int())
    # chunked, overlapped all-gather (what bench.py --gpus N runs)
    def score_chunk(lo, hi, out):
        out.copy_(orc.sim_scores(d["q"][lo:hi], g, mask)[0])
    ov = ddist.OverlappedShardScorer(score_chunk, nq, s, 3, "cpu")
    ov.step()
    ov.step()                                            # buffers are reusable across steps
    full_ov = ov.assemble(nv)
    # gradient bucket
    torch.manual_seed(rank)
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    for p in ps:
        p.grad = torch.full_like(p, float(rank + 1))
    b = ddist.FlatGradBucket(ps)
    b.all_reduce_mean()
    if rank == 0:
        ret["full"], ret["ranks"], ret["grad"] = full.numpy(), ranks.numpy(), [p.grad.clone().numpy() for p in ps]
        ret["full_ov"] = full_ov.numpy()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nv,nq", [(10, 12), (7, 9)])       # 7 videos over 2 ranks: uneven shards + padding
def test_sharded_eval_two_ranks(nv, nq):
    world, port = 2, _free_port()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, nv, nq, ret), nprocs=world, join=True)
    d = synth.make_gallery(77, nq, nv, 16, 3, sigma=3.0)
    ref = orc.sim_scores(d["q"], d["g"], d["mask"])[0].numpy()
    np.testing.assert_allclose(ret["full"], ref, rtol=0, atol=1e-6)           # sharded == unsharded
    np.testing.assert_allclose(ret["full_ov"], ref, rtol=0, atol=1e-6)        # chunked + overlapped gather too
    gts = {q: [int(d["gt"][q])] for q in range(nq)}
    assert (ret["ranks"] == orc.gt_ranks(-ref, gts)).all()                    # gather-free ranks are exact
    for gr in ret["grad"]:
        np.testing.assert_allclose(gr, 1.5)                                   # mean of ranks' grads (1, 2)


def test_shard_range_covers_everything():
    from dldkd_amd import dist as ddist
    for nv in (1, 7, 8, 21793, 4917):
        for world in (1, 2, 4, 8):
            cover = []
            for r in range(world):
                lo, hi, s = ddist.shard_range(nv, r, world)
                assert 0 <= hi - lo <= s
                cover += list(range(lo, hi))
            assert cover == list(range(nv))
