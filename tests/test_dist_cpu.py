"""CPU, gloo, world_size 2: the N>1 plumbing of the path that is SHIPPED - gallery sharding, the overlapped per-range
all-gather assembly (OverlappedShardScorer driven by an injected CPU backend: the oracle), gather-free ranking, and the
data-parallel half of the training step (train.train_step -> dist.sync_gradients on optimization.FlatParams, with DIFFERENT
batches per rank, a DistributedSampler-cut dataset and the parameter broadcast).  No GPU here: scorer, count and the optimizer
update are injected; the collectives, shard arithmetic, padding, divisor and ordering are the code under test."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import dldkd_oracle as orc
import synth


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _setup(rank, world, port):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in ("dl-dkd_amd", "oracle", "tests/golden"):
        sys.path.insert(0, os.path.join(root, p))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)


class _OracleBackend:
    """CPU stand-in for dist.HipShardBackend: launch() scores everything (the oracle) and marks every range complete;
    wait_range() checks the counter like the stream wait would."""

    def __init__(self, q, g, mask, n_ranges):
        self.q, self.g, self.mask, self.arrivals, self.n_ranges = q, g, mask, 3, n_ranges
        self.launches = 0

    def launch(self, done):
        assert int(done.abs().sum()) == 0                   # the scorer zeroes the counters before every launch
        self.full = orc.sim_scores(self.q, self.g, self.mask)[0]
        done[:self.n_ranges] = self.arrivals
        self.launches += 1

    def wait_range(self, done, r):
        assert int(done[r]) >= self.arrivals

    def finish_range(self, r, lo, hi, out):
        out.copy_(self.full[lo:hi])


def _eval_worker(rank, world, port, nv, nq, ret):
    _setup(rank, world, port)
    from dldkd_amd import dist as ddist
    d = synth.make_gallery(77, nq, nv, 16, 3, sigma=3.0)
    lo, hi, s = ddist.shard_range(nv, rank, world)
    g, mask = d["g"][lo:hi], d["mask"][lo:hi]
    if hi - lo < s:                                   # pad the short shard like bench.py does
        pad = s - (hi - lo)
        g = torch.cat([g, torch.zeros(pad, 16, 384)]); mask = torch.cat([mask, torch.zeros(pad, 16)]); mask[hi - lo:, 0] = 1
    local = orc.sim_scores(d["q"], g, mask)[0]        # (Nq, S): the injected scorer
    gt = d["gt"].clone()
    gt[0] = -1                                        # a caption whose video is not in the gallery: rank nv + 1
    ranks = ddist.sharded_gt_ranks(local, gt, nv, count_fn=lambda sc, thr, n: (~(sc[:, :n] <= thr[:, None])).sum(1).int())
    # a NaN ground-truth score ranks last on every rank
    local_nan = local.clone()
    if lo <= int(d["gt"][1]) < hi:
        local_nan[1, int(d["gt"][1]) - lo] = float("nan")
    ranks_nan = ddist.sharded_gt_ranks(local_nan, d["gt"], nv, count_fn=lambda sc, thr, n: (~(sc[:, :n] <= thr[:, None])).sum(1).int())
    # one launch, per-range finish + all-gather (what bench.py --gpus N runs)
    bounds = ddist.query_ranges(nq, 3, 4)
    be = _OracleBackend(d["q"], g, mask, len(bounds))
    ov = ddist.OverlappedShardScorer(be, bounds, s, "cpu")
    ov.step()
    ov.step()                                            # buffers and counters are reusable across steps
    full_ov = ov.assemble(nv)
    if rank == 0:
        ret["ranks"], ret["ranks_nan"], ret["full_ov"], ret["launches"] = ranks.numpy(), ranks_nan.numpy(), full_ov.numpy(), be.launches
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nv,nq", [(10, 12), (7, 9)])       # 7 videos over 2 ranks: uneven shards + padding
def test_sharded_eval_two_ranks(nv, nq):
    world, port = 2, _free_port()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_eval_worker, args=(world, port, nv, nq, ret), nprocs=world, join=True)
    d = synth.make_gallery(77, nq, nv, 16, 3, sigma=3.0)
    ref = orc.sim_scores(d["q"], d["g"], d["mask"])[0].numpy()
    np.testing.assert_allclose(ret["full_ov"], ref, rtol=0, atol=1e-6)        # per-range gathered == unsharded
    assert ret["launches"] == 2                                               # ONE scorer launch per step
    gts = {q: [int(d["gt"][q])] for q in range(nq)}
    want = orc.gt_ranks(-ref, gts)
    assert (ret["ranks"][1:] == want[1:]).all() and ret["ranks"][0] == nv + 1  # gather-free ranks are exact
    assert ret["ranks_nan"][1] == nv + 1 and (np.delete(ret["ranks_nan"], 1) == np.delete(want, 1)).all()


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
    assert ddist.query_ranges(17505, 4, 4384) == [(0, 4384), (4384, 8768), (8768, 13152), (13152, 17505)]


# ------------------------------------------------------------------------------------------ training, data parallel
class _ToyModel(torch.nn.Module):
    """Stands in for DLDKD in the CPU test of the shipped train_step: same calling convention (batch dict -> (loss, dict)),
    several parameter shapes, one parameter that never receives a gradient."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(6, 5)
        self.b = torch.nn.Linear(5, 1, bias=False)
        self.unused = torch.nn.Parameter(torch.ones(3))

    def forward(self, batch):
        y = self.b(torch.tanh(self.a(batch["x"]))).squeeze(-1)
        loss = ((y - batch["y"]) ** 2).mean()
        return loss, {"loss_overall": float(loss.detach())}


class _SGDOnFlat:
    """Injected optimizer: the FlatParams machinery of BertAdam (shipped) with a plain SGD update (the fused HIP update needs a
    GPU).  zero_grad / step follow BertAdam's: drop the grads, rebind, update the flat buffer."""

    def __init__(self, params, lr):
        from dldkd_amd.optimization import FlatParams
        self.fp, self.lr = FlatParams(list(params)), lr

    def zero_grad(self):
        self.fp.drop_grads()

    def step(self):
        had = self.fp.rebind_grads()
        self.fp.flat.sub_(self.lr * self.fp.grad)
        return had


def _data(n=24):
    g = torch.Generator().manual_seed(5)
    return torch.randn(n, 6, generator=g), torch.randn(n, generator=g)


def _train_worker(rank, world, port, ret):
    _setup(rank, world, port)
    from torch.utils.data import TensorDataset
    from dldkd_amd import dist as ddist
    from dldkd_amd import train as T
    torch.manual_seed(100 + rank)                       # replicas are BORN different ...
    model = _ToyModel()
    opt_ = _SGDOnFlat(model.parameters(), lr=0.1)
    ddist.broadcast_parameters(opt_.fp)                  # ... and start from rank 0's weights
    start = opt_.fp.flat.clone()
    x, y = _data()
    ds = TensorDataset(x, y)
    cfg = types.SimpleNamespace(bsz=4, pin_memory=False, num_workers=0, seed=3, grad_clip=-1)
    # per-rank data: the DistributedSampler of make_train_loader (collate replaced: toy items are (x, y) pairs)
    loader = T.make_train_loader(ds, cfg, rank, world)
    seen = []
    hist = []
    for epoch in range(2):
        loader.sampler.set_epoch(epoch)
        idx = list(iter(loader.sampler))
        seen.append(idx)
        for i in range(0, len(idx), cfg.bsz):
            b = idx[i:i + cfg.bsz]
            batch = {"x": x[b], "y": y[b]}
            _, ld = T.train_step(model, batch, opt_, cfg)          # the shipped step: fwd, bwd, sync_gradients, update
            hist.append((b, ld["loss_overall"], opt_.fp.grad.clone()))
    ret[rank] = dict(start=start.numpy(), final=opt_.fp.flat.clone().numpy(), seen=seen,
                     grads=[h[2].numpy() for h in hist], batches=[h[0] for h in hist],
                     unused=model.unused.detach().numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_shipped_train_step_two_ranks_different_batches():
    """What the one-rank RCCL test could not catch (mean over one rank is the identity): the divisor, a missing reduce, replicas
    that drift.  Two ranks, different batches; the averaged gradient must equal the mean of the per-rank gradients computed
    here in ONE process, and both replicas must hold identical parameters after every step."""
    world, port = 2, _free_port()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_train_worker, args=(world, port, ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    np.testing.assert_array_equal(r0["start"], r1["start"])                   # broadcast from rank 0
    np.testing.assert_array_equal(r0["final"], r1["final"])                   # replicas never drift
    np.testing.assert_array_equal(r0["unused"], np.ones(3, np.float32))
    for e in range(2):                                                        # the sampler cuts each epoch disjointly
        a, b = set(r0["seen"][e]), set(r1["seen"][e])
        assert not (a & b) and a | b == set(range(24))
    assert r0["seen"][0] != r0["seen"][1]                                     # set_epoch reshuffles
    # replay in one process: same start, per-rank batches, gradient = MEAN of the two ranks' gradients
    torch.manual_seed(100)
    model = _ToyModel()
    from dldkd_amd.optimization import FlatParams
    fp = FlatParams(list(model.parameters()))
    np.testing.assert_array_equal(fp.flat.numpy(), r0["start"])
    x, y = _data()
    for step, (b0, b1) in enumerate(zip(r0["batches"], r1["batches"])):
        gs = []
        for b in (b0, b1):
            fp.drop_grads()
            loss, _ = model({"x": x[b], "y": y[b]})
            loss.backward()
            fp.rebind_grads()
            gs.append(fp.grad.clone())
        mean = (gs[0] + gs[1]) / 2
        assert not torch.equal(gs[0], gs[1])                                  # the batches really differ
        np.testing.assert_allclose(r0["grads"][step], mean.numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(r0["grads"][step], r1["grads"][step])
        fp.flat.sub_(0.1 * mean)
    np.testing.assert_allclose(fp.flat.numpy(), r0["final"], rtol=1e-5, atol=1e-6)


class _ToyPhased(_ToyModel):
    """The toy with the phased-backward protocol of DLDKD.forward_phased: the loss reaches `a` only through the hidden
    activation (the tap); `b` sits behind the tap (a head phase, tap None); `unused` is in no bucket (the rest range)."""

    def grad_buckets(self):
        return [list(self.a.parameters()), list(self.b.parameters())]

    def forward_phased(self, batch):
        h = torch.tanh(self.a(batch["x"]))
        y = self.b(h).squeeze(-1)
        loss = ((y - batch["y"]) ** 2).mean()
        return loss, {"loss_overall": float(loss.detach())}, [(h, list(self.a.parameters())), (None, list(self.b.parameters()))]


class _SGDBucketed(_SGDOnFlat):
    def __init__(self, params, lr, buckets):
        from dldkd_amd.optimization import FlatParams
        self.fp, self.lr = FlatParams(list(params), buckets), lr


def _bucketed_worker(rank, world, port, ret):
    _setup(rank, world, port)
    from dldkd_amd import train as T
    x, y = _data()
    cfg = types.SimpleNamespace(grad_clip=-1)
    out = {}
    for kind in ("single", "bucketed"):
        torch.manual_seed(7)                                # same start on both ranks and in both variants
        model = _ToyPhased() if kind == "bucketed" else _ToyModel()
        opt_ = _SGDBucketed(model.parameters(), 0.1, model.grad_buckets()) if kind == "bucketed" else _SGDOnFlat(model.parameters(), 0.1)
        if kind == "bucketed":
            # the layout really is bucket by bucket: a | b | rest, contiguous and disjoint
            assert len(opt_.fp.bucket_ranges) == 3 and opt_.fp.bucket_ranges[0][0] == 0
            assert all(opt_.fp.bucket_ranges[i][1] == opt_.fp.bucket_ranges[i + 1][0] for i in range(2))
        grads = []
        for step in range(4):
            b = [(8 * step + 4 * rank + i) % 24 for i in range(4)]      # different batches per rank
            T.train_step(model, {"x": x[b], "y": y[b]}, opt_, cfg)
            grads.append({n: p.grad.clone().numpy() for n, p in model.named_parameters()})
        out[kind] = dict(final={n: p.detach().clone().numpy() for n, p in model.named_parameters()}, grads=grads)
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_overlap_equals_single_bucket_two_ranks():
    """The data-parallel step with the backward pass run in phases and one all-reduce per gradient bucket (train.
    backward_in_phases + dist.BucketedGradSync over the bucket-contiguous FlatParams layout) against the single all-reduce of the
    whole buffer: same averaged gradients and same parameters, bit for bit, on both ranks, the gradient-less parameter
    untouched."""
    world, port = 2, _free_port()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_bucketed_worker, args=(world, port, ret), nprocs=world, join=True)
    for rank in (0, 1):
        one, many = ret[rank]["single"], ret[rank]["bucketed"]
        for n in one["final"]:
            np.testing.assert_array_equal(one["final"][n], many["final"][n])
        for ga, gb in zip(one["grads"], many["grads"]):
            for n in ga:
                np.testing.assert_array_equal(ga[n], gb[n])
        np.testing.assert_array_equal(many["final"]["unused"], np.ones(3, np.float32))
    for n in ret[0]["bucketed"]["final"]:
        np.testing.assert_array_equal(ret[0]["bucketed"]["final"][n], ret[1]["bucketed"]["final"][n])
    assert not np.array_equal(ret[0]["bucketed"]["grads"][0]["a.weight"], np.zeros_like(ret[0]["bucketed"]["grads"][0]["a.weight"]))


def test_flat_params_bucket_layout_keeps_tensor_order_of_per_tensor_arrays():
    from dldkd_amd.optimization import FlatParams, CHUNK
    ps = [torch.nn.Parameter(torch.full((n,), float(i))) for i, n in enumerate((5, 300, 7, 256, 1))]
    fp = FlatParams(ps, [[ps[3], ps[1]], [ps[4]]])
    assert fp.bucket_params == [[1, 3], [4], [0, 2]]
    assert fp.bucket_ranges == [(0, 3 * CHUNK), (3 * CHUNK, 4 * CHUNK), (4 * CHUNK, 6 * CHUNK)]
    assert fp.t_numel.tolist() == [5, 300, 7, 256, 1] and fp.t_start.tolist() == [4 * CHUNK, 0, 5 * CHUNK, 2 * CHUNK, 3 * CHUNK]
    assert fp.chunk_tensor.tolist() == [1, 1, 3, 4, 0, 2]
    for i, p in enumerate(ps):                              # values survive the move, storage is the flat buffer
        assert torch.equal(p.data, torch.full_like(p.data, float(i)))
        assert p.data.data_ptr() == fp.flat[fp.t_start[i]:].data_ptr()
    with pytest.raises(ValueError):
        FlatParams(ps, [[ps[0]], [ps[0]]])


# ------------------------------------------------------------------------------------------------ bench.py --gpus N setup
_BENCH_CFG = dict(name="T", nq=57, nv=150, L=8, len_lo=2, seed=9, sigma=(0.5, 0.7), workload="test")


def _bench_setup_worker(rank, world, port, ret):
    _setup(rank, world, port)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    cfg = _BENCH_CFG
    shard = (cfg["nv"] + world - 1) // world
    lo, hi = min(rank * shard, cfg["nv"]), min((rank + 1) * shard, cfg["nv"])
    from dldkd_amd import comm as dcomm
    gs, mask, lens, qs, gt = bench.synth_shard("cpu", cfg, lo, hi, world, dcomm.current())
    ret[rank] = dict(lo=lo, hi=hi, gs=[g.clone() for g in gs], lens=lens.clone(), qs=[q.clone() for q in qs], gt=gt.clone())
    dist.destroy_process_group()


def test_bench_multi_rank_setup_is_rank_invariant():
    """bench.py --gpus N: every rank holds the SAME queries (planted clip all-reduced from the rank that owns
    the ground-truth video) and the union of the shards IS the one-GPU gallery (per-block seeds, independent of the cut)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    world, port = 2, _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_bench_setup_worker, args=(world, port, ret), nprocs=world, join=True)
    cfg = _BENCH_CFG
    gs1, mask1, lens1, qs1, gt1 = bench.synth_shard("cpu", cfg, 0, cfg["nv"])
    a, b = ret[0], ret[1]
    assert (a["lo"], a["hi"], b["lo"], b["hi"]) == (0, 75, 75, 150)
    for br in range(2):
        assert torch.equal(a["qs"][br], b["qs"][br])                       # identical query set on both ranks ...
        assert torch.equal(a["qs"][br], qs1[br])                           # ... and it is the one-GPU query set
        assert torch.equal(torch.cat([a["gs"][br], b["gs"][br]]), gs1[br])  # shards = the one-GPU gallery cut by video
    assert torch.equal(torch.cat([a["lens"], b["lens"]]), lens1) and torch.equal(a["gt"], gt1)
    # an odd cut (not on a block boundary) gives the same videos too
    g_mid, l_mid = bench.synth_videos("cpu", cfg, 37, 101)
    assert torch.equal(g_mid[1], gs1[1][37:101]) and torch.equal(l_mid, lens1[37:101])
    # the planted signal survives: a query's own video holds its best-matching clip far more often than chance
    import torch.nn.functional as F
    sims = torch.einsum("qd,vld->qvl", F.normalize(qs1[0], dim=-1), F.normalize(gs1[0], dim=-1))
    sims = sims.masked_fill(mask1[None] == 0, -2.0).amax(-1)
    assert (sims.argmax(1) == gt1).float().mean().item() > 0.5


# ------------------------------------------------------------------------------------------------ sharded ranks from the partial planes
def _ranks_worker(rank, world, port, ret):
    _setup(rank, world, port)
    from dldkd_amd import dist as ddist
    g = torch.Generator().manual_seed(3)
    nq, nv = 23, 11
    s = torch.randn(3, nq, nv, generator=g)                         # three score kinds, the whole gallery
    gt = {q: [int(v) for v in torch.randperm(nv, generator=g)[:1 + q % 3]] for q in range(nq)}
    del gt[4]                                                       # a caption whose video is not in the gallery
    s[:, 7, gt[7][0]] = float("nan")                                # NaN score of a first-listed GT video
    bad = torch.zeros(nq, dtype=torch.bool); bad[9] = True          # a flagged (NaN / Inf) query
    lo, hi, _ = ddist.shard_range(nv, rank, world)
    ptr, idx, first, has = ddist.local_gt_csr(gt, nq, lo, hi)
    loc = s[:, :, lo:hi]

    def thr_fn():                                                   # what rank_part_thr_kernel computes in shard mode
        thr = torch.full((3, 2, nq), float("-inf")); flag = torch.zeros(3, 2, nq)
        for q in range(nq):
            mine = idx[ptr[q]:ptr[q + 1]]
            for k in range(3):
                vals = loc[k, q, mine] if len(mine) else torch.empty(0)
                ok = vals[~torch.isnan(vals)]
                if len(ok):
                    thr[k, 0, q] = ok.max()
                if first[q]:
                    f = loc[k, q, gt[q][0] - lo]
                    if torch.isnan(f):
                        flag[k, 1, q] = 1.0
                    else:
                        thr[k, 1, q] = f
        return thr, flag

    def count_fn(thr):
        return (~(loc[:, None, :, :] <= thr[:, :, :, None])).sum(-1).int()          # (3, 2, nq)
    ranks = ddist.sharded_ranks_from_partials(thr_fn, count_fn, torch.from_numpy(has), bad, nv)
    ret[rank] = ranks.numpy()
    if rank == 0:
        want = np.zeros((3, 2, nq), np.int64)
        for q in range(nq):
            for k in range(3):
                if q not in gt or bad[q]:
                    want[k, :, q] = nv + 1
                    continue
                row = s[k, q]
                vals = row[gt[q]]
                ok = vals[~torch.isnan(vals)]
                best = ok.max() if len(ok) else torch.tensor(float("-inf"))
                want[k, 0, q] = min(1 + int((~(row <= best)).sum()), nv + 1)
                f = row[gt[q][0]]
                want[k, 1, q] = nv + 1 if torch.isnan(f) else min(1 + int((~(row <= f)).sum()), nv + 1)
        ret["want"] = want
    dist.destroy_process_group()


def test_sharded_ranks_from_partials_two_ranks():
    """dist.sharded_ranks_from_partials (eval_epoch_sharded's ranking: thresholds over local GT videos -> all-reduce MAX ->
    local counts -> all-reduce SUM, no score matrix) equals ranking the whole matrix: several GT videos per query spread over
    both shards, a query without ground truth, a flagged query, a NaN first-GT score."""
    world, port = 2, _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_ranks_worker, args=(world, port, ret), nprocs=world, join=True)
    assert (ret[0] == ret[1]).all()                                  # the same ranks on every rank
    assert (ret[0] == ret["want"]).all(), (ret[0][0], ret["want"][0])


# ------------------------------------------------------------------------------------------------ a peer that stops stepping
def _stall_worker(rank, world, port, ret):
    import time
    os.environ["DLDKD_COMM_DEADLINE_S"] = "2"
    _setup(rank, world, port)
    from dldkd_amd import comm as dcomm
    from dldkd_amd import dist as ddist
    from dldkd_amd import train as T
    torch.manual_seed(100 + rank)
    model = _ToyModel()
    opt_ = _SGDOnFlat(model.parameters(), lr=0.1)
    ddist.broadcast_parameters(opt_.fp)
    x, y = _data()
    cfg = types.SimpleNamespace(bsz=4, grad_clip=-1)
    batch = {"x": x[4 * rank:4 * rank + 4], "y": y[4 * rank:4 * rank + 4]}
    T.train_step(model, batch, opt_, cfg)                 # one healthy step on both ranks
    if rank == 1:
        time.sleep(8.0)                                   # alive, connected, but no longer stepping (a diverged / stuck peer)
        ret[rank] = dict(raised=None, seconds=None)
        return
    t0 = time.monotonic()
    try:
        T.train_step(model, batch, opt_, cfg)
        raised = None
    except dcomm.CommTimeout as ex:
        raised = str(ex)
    ret[rank] = dict(raised=raised, seconds=time.monotonic() - t0)


def test_a_rank_that_stops_stepping_makes_the_other_raise_within_the_deadline():
    """with the process-group watchdog gone nothing bounded a wait on a dead or diverged peer.  Every blocking
    point of the multi-rank path is deadline-bounded in the calling thread (comm.Comm.host_wait / the polled collectives of
    TorchGroupComm): rank 1 stops stepping, rank 0's next train_step raises CommTimeout at DLDKD_COMM_DEADLINE_S = 2 s, not never."""
    world, port = 2, _free_port()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_stall_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0]["raised"] is not None and "deadline" in ret[0]["raised"] and "rank 0 of 2" in ret[0]["raised"]
    assert 1.5 <= ret[0]["seconds"] < 6.0


# ------------------------------------------------------------------------------------------------ rank-invariant had-flag check
def _had_worker(rank, world, port, ret):
    _setup(rank, world, port)
    from dldkd_amd import dist as ddist
    from dldkd_amd.optimization import FlatParams
    torch.manual_seed(3)
    model = _ToyModel()
    fp = FlatParams(list(model.parameters()))
    n = len(fp.params)
    same = tuple(True for _ in range(n))
    # (1) identical sets: calls 0 and 1 check, later ones do not - on BOTH ranks alike, whatever the rank-local history is
    events = []
    for step in range(5):
        fp.grad.fill_(float(rank + 1))
        ddist.sync_gradients(fp, had=same)
        events.append(float(fp.grad[0]))
    # (2) rank 1's set changes at a step where no check is due: nothing is issued by either rank (no unpaired collective: the
    # gradient all-reduce that follows still pairs up and returns the mean)
    mine = tuple(i != 0 or rank == 0 for i in range(n))
    fp.grad.fill_(float(rank + 1))
    ddist.sync_gradients(fp, had=mine)
    events.append(float(fp.grad[0]))
    # (3) at the next due call the divergence is caught on BOTH ranks, by a collective both take part in
    fp._had_calls = ddist.HAD_CHECK_EVERY
    try:
        ddist.sync_gradients(fp, had=mine)
        caught = False
    except RuntimeError as ex:
        caught = "disagree" in str(ex)
    ret[rank] = dict(events=events, caught=caught, calls=fp._had_calls)
    dist.barrier()
    dist.destroy_process_group()


def test_had_flag_check_is_collective_on_every_rank():
    """the check of the ranks' "had a gradient" sets must not be gated on rank-local state - a rank whose set
    changed would issue collectives the others do not.  It runs on a call-count cadence that is the same on every rank; a
    divergence is reported by RuntimeError on every rank at the next due call, never by a hang or a corrupted gradient buffer."""
    world, port = 2, _free_port()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_had_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(2):
        assert ret[r]["events"] == [1.5] * 6                       # mean of 1 and 2 at every step, including the diverged one
        assert ret[r]["caught"] is True
    assert ret[0]["calls"] == ret[1]["calls"]


# ------------------------------------------------------------------------------------------------ RCCL rendezvous id over the env:// store
def _store_worker(rank, world, port, ret):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "dl-dkd_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from dldkd_amd import comm as dcomm
    store, r, w = dcomm._env_store(rank, world, timeout_s=60)
    if r == 0:
        uid = dcomm.RcclComm.unique_id()                 # host-only call of librccl: works without a GPU
        store.set("id", uid)
    else:
        uid = bytes(store.get("id"))
    store.add("seen", 1)
    while int(store.add("seen", 0)) < w:                 # rank 0 hosts the store: keep it alive until everyone has read
        pass
    ret[rank] = (r, w, uid)


def test_rccl_unique_id_travels_over_the_env_store():
    """comm.init_rccl_from_env's host half without a GPU: rank 0 draws the 128-byte id through the C ABI (dldkd_comm_unique_id) and
    every rank reads the same bytes from the env:// TCP store."""
    world, port = 2, _free_port()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_store_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0][:2] == (0, 2) and ret[1][:2] == (1, 2)
    assert len(ret[0][2]) == 128 and ret[0][2] == ret[1][2] and any(ret[0][2])
