"""GPU: the eval driver end to end (encode gallery -> score -> rank) against the reference's own
eval_epoch outputs (golden G5) and the oracle."""
import types

import numpy as np
import pytest
import torch

import dldkd_oracle as orc
import synth
from test_encoder_gpu import _model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _opt():
    # eval_precision="parity": these tests pin eval_epoch to the reference's outputs through the fp32-grade towers; the
    # throughput default of eval_epoch is gated in tests/test_rk_gate_gpu.py
    return types.SimpleNamespace(eval_context_bsz=25, eval_query_bsz=50, num_workers=0, pin_memory=False,
                                 device=torch.device(DEV), double_branch=True, eval_precision="parity")


def test_rank_kernel_vs_oracle():
    from dldkd_amd import eval as ev
    g = torch.Generator().manual_seed(5)
    for nq, nv in ((1, 1), (37, 5), (64, 1001), (50, 21793)):
        s = torch.randn(nq, nv, generator=g)
        gts = {q: sorted(set(torch.randint(0, nv, (1 + q % 3,), generator=g).tolist())) for q in range(nq)}
        rb, rf = ev.gt_ranks_gpu(s.to(DEV), gts)
        ref = orc.gt_ranks(-s.numpy(), gts)
        assert (rb.cpu().numpy() == ref).all()
        first = np.array([1 + int((s[q] > s[q, gts[q][0]]).sum()) for q in range(nq)])
        assert (rf.cpu().numpy() == first).all()
        assert ev.eval_q2m(-s.numpy(), gts) == pytest.approx(orc.eval_q2m(-s.numpy(), gts))
        assert ev.t2v_map(-s.numpy(), gts) == pytest.approx(orc.t2v_map(-s.numpy(), gts))


def test_get_gt_matches_oracle():
    from dldkd_amd import eval as ev
    vm = [f"v{i}" for i in range(7)]
    qm = ["v3#a", "v0#b#c", "zz#q", "v3#d", "v6#e"]
    assert ev.get_gt(vm, qm) == orc.get_gt(vm, qm)


def test_eval_epoch_vs_golden_g5(golden_dir):
    from dldkd_amd import eval as ev
    g = np.load(f"{golden_dir}/g5_eval_epoch.npz")
    m = _model(3072, 768, synth.make_params(51, 3072, 768))
    vids, txts = synth.make_eval_sets(5, nv=64, caps=3, dv=3072, dq=768)
    opt = _opt()
    with torch.no_grad():
        ctx = ev.compute_context_info(m, synth.ListDataset(list(vids)), opt)
        inh, exp, _, qmetas = ev.compute_query2ctx_info(m, synth.ListDataset(list(txts)), opt, ctx)
        sumr = ev.eval_epoch(m, synth.ListDataset(list(vids)), synth.ListDataset(list(txts)), opt)
    assert list(qmetas) == list(g["query_metas"])                 # per-batch length-sorted row order
    assert list(ctx["video_metas"]) == list(g["video_metas"])
    assert (ctx["video_mask"].cpu().numpy() == g["video_mask"]).all()
    samp = ctx["inher_frame_feat"][::7, ::3, ::16].cpu().numpy()
    assert np.abs(samp - g["gallery_inh_sample"]).max() < 3e-5     # fp32 towers
    assert np.abs(inh - g["inh"]).max() < 6e-3 and np.abs(exp - g["exp"]).max() < 6e-3   # bf16 scorer
    # R@K of the reference on these inputs; random-init weights give near-ties, so allow one query (of 192)
    # to move across a cut: 100/192 = 0.52
    _, t2v = ev.get_gt(ctx["video_metas"], qmetas)
    fused = 0.7 * inh + 0.3 * exp
    ours = ev.eval_q2m(-fused, t2v)
    for a, b in zip(ours[:4], g["perf_fused"][:4]):
        assert abs(a - b) <= 0.53, (ours, g["perf_fused"])
    assert abs(sumr - float(g["sumr"])) <= 1.6
    # and exactly equal to ranking OUR scores with the oracle's ranking code
    assert ours == pytest.approx(orc.eval_q2m(-fused, t2v))


def test_get_pred_from_raw_query():
    from dldkd_amd import eval as ev
    from dldkd_amd.data import collate_text_val
    m = _model(1024, 1024, synth.make_params(9, 1024, 1024))
    vids, txts = synth.make_eval_sets(6, nv=10, caps=2, dv=1024, dq=1024)
    opt = _opt()
    with torch.no_grad():
        ctx = ev.compute_context_info(m, synth.ListDataset(list(vids)), opt)
        words, mask, _, _ = collate_text_val(list(txts))
        s0, s1 = m.get_pred_from_raw_query(words.to(DEV), mask.to(DEV), ctx)
    p = {k: v.cpu() for k, v in m.state_dict().items()}
    qi, qe = orc.encode_query(p, words, mask)
    ref0 = orc.sim_scores(qi, ctx["inher_frame_feat"].cpu(), ctx["video_mask"].cpu())[0]
    ref1 = orc.sim_scores(qe, ctx["explore_frame_feat"].cpu(), ctx["video_mask"].cpu())[0]
    assert (s0.cpu() - ref0).abs().max() < 6e-3 and (s1.cpu() - ref1).abs().max() < 6e-3


def test_eval_epoch_sharded_equals_unsharded():
    """One-rank RCCL communicator: the sharded driver (gather-free ranking) returns the unsharded SumR."""
    from dldkd_amd import comm as dcomm
    from dldkd_amd import eval as ev
    m = _model(1024, 1024, synth.make_params(13, 1024, 1024))
    vids, txts = synth.make_eval_sets(8, nv=37, caps=2, dv=1024, dq=1024)
    opt = _opt()
    with torch.no_grad():
        ref = ev.eval_epoch(m, synth.ListDataset(list(vids)), synth.ListDataset(list(txts)), opt)
        a = ev.eval_epoch_sharded(m, synth.ListDataset(list(vids)), synth.ListDataset(list(txts)), opt)   # no communicator
        assert a == pytest.approx(ref)
        c = dcomm.install(dcomm.RcclComm(1, 0, dcomm.RcclComm.unique_id(), torch.device(DEV)))
        try:
            b = ev.eval_epoch_sharded(m, synth.ListDataset(list(vids)), synth.ListDataset(list(txts)), opt)
        finally:
            dcomm.install(None)
            c.destroy()
    assert b == pytest.approx(ref)


def test_gallery_packer_streaming_equals_one_shot():
    """Chunks with their own (shorter) padded length land in the blob exactly as the one-shot packer puts them."""
    from dldkd_amd import scoring
    g = torch.Generator().manual_seed(21)
    nv, L = 23, 50
    lens = torch.randint(1, L + 1, (nv,), generator=g)
    lens[4] = L
    mask = (torch.arange(L)[None] < lens[:, None]).float()
    gs = [(torch.randn(nv, L, 384, generator=g) * mask[..., None]).to(DEV) for _ in range(2)]
    one = scoring.pack_gallery(gs, mask.to(DEV))
    pk = scoring.GalleryPacker(nv, L, 2, torch.device(DEV))
    for lo, hi in ((0, 7), (7, 8), (8, 23)):
        lc = int(lens[lo:hi].max())                              # collate pads each batch to ITS max length
        pk.add([x[lo:hi, :lc].contiguous() for x in gs], mask[lo:hi, :lc].contiguous().to(DEV))
    st = pk.finish()
    assert torch.equal(st.lens, one.lens) and torch.equal(st.order, one.order) and torch.equal(st.inv_order, one.inv_order)
    for a, b in zip(st.blobs, one.blobs):
        assert torch.equal(a, b)
    with pytest.raises(Exception):
        scoring.GalleryPacker(3, L, 1, torch.device(DEV)).finish()          # nothing packed yet
    with pytest.raises(Exception):
        scoring.GalleryPacker(3, 129, 1, torch.device(DEV))


def test_streaming_context_and_query_super_batches_equal_per_batch_path():
    from dldkd_amd import eval as ev
    m = _model(1024, 1024, synth.make_params(17, 1024, 1024))
    vids, txts = synth.make_eval_sets(9, nv=41, caps=3, dv=1024, dq=1024)
    opt = _opt()
    opt.eval_query_bsz = 7
    with torch.no_grad():
        full = ev.compute_context_info(m, synth.ListDataset(list(vids)), opt)
        stream = ev.compute_context_info(m, synth.ListDataset(list(vids)), opt, keep_frame_feats=False)
        assert stream["inher_frame_feat"] is None and torch.equal(stream["video_mask"], full["video_mask"])
        assert torch.equal(stream["_packed"].lens, full["_packed"].lens)
        f1, a1, b1, metas1 = ev.score_queries(m, synth.ListDataset(list(txts)), opt, full)
        f2, a2, b2, metas2 = ev.score_queries(m, synth.ListDataset(list(txts)), opt, stream)
        assert metas1 == metas2
        assert torch.equal(f1, f2) and torch.equal(a1, a2) and torch.equal(b1, b2)
        # super-batched query encoding (here: all 123 queries in one pass) vs one encode per loader batch
        old, ev.QUERY_SUPER_BATCH = ev.QUERY_SUPER_BATCH, 1
        try:
            _, qs_small = ev._encode_all_queries(m, synth.ListDataset(list(txts)), opt)
        finally:
            ev.QUERY_SUPER_BATCH = old
        _, qs_big = ev._encode_all_queries(m, synth.ListDataset(list(txts)), opt)
        for x, y in zip(qs_small, qs_big):
            assert x.shape == y.shape and (x - y).abs().max().item() < 2e-6


def test_eval_epoch_sharded_edge_cases(recwarn, rccl_comm):
    """ids come from the dataset's `video_ids` attribute (the reference's VisDataSet4DLDKD has it, data_provider.py:270-275):
    no feature is read to build the ground truth; a caption whose video is missing from the gallery ranks nv + 1 like in the
    unsharded path; an empty gallery shard (more ranks than videos) scores nothing and still returns."""
    from dldkd_amd import eval as ev
    m = _model(1024, 1024, synth.make_params(13, 1024, 1024))
    vids, txts = synth.make_eval_sets(8, nv=21, caps=2, dv=1024, dq=1024)
    txts = list(txts) + [(txts[0][0], len(txts), "no_such_video#enc#0")]

    class CountingGallery(synth.ListDataset):
        reads = 0

        def __init__(self, items):
            super().__init__(items)
            self.video_ids = [it[2] for it in items]

        def __getitem__(self, i):
            CountingGallery.reads += 1
            return self.items[i]
    opt = _opt()
    with torch.no_grad():
        ref = ev.eval_epoch(m, synth.ListDataset(list(vids)), synth.ListDataset(list(txts)), opt)
        try:
            gal = CountingGallery(list(vids))
            got = ev.eval_epoch_sharded(m, gal, synth.ListDataset(list(txts)), opt)
            assert CountingGallery.reads == len(vids)              # each video read once (to encode it), none for the ids
            assert ev.gallery_ids(gal) == [v[2] for v in vids]
            # empty shard: what rank r >= n_videos sees
            ctx = ev.compute_context_info(m, synth.ListDataset([]), opt, keep_frame_feats=False)
            fused, s0, s1, _ = ev.score_queries(m, synth.ListDataset(list(txts)), opt, ctx)
            assert fused.shape == (len(txts), 0) and ctx["_packed"].nv == 0
        finally:
            torch.cuda.synchronize()
    assert got == pytest.approx(ref)
    assert not [w for w in recwarn.list if "video_ids" in str(w.message)]


@pytest.mark.parametrize("nq,nv,L,nb", [(10895, 2000, 128, 2), (333, 77, 40, 2), (64, 1, 9, 2), (50, 600, 128, 1)])
def test_ranks_from_partials_equal_ranks_from_matrices(nq, nv, L, nb):
    """scoring.rank_partials (eval_epoch's path: no score matrix written) against the matrix path
    (simpool_finish + rank_gt per matrix): identical integer ranks for all three score kinds and both rank flavours, with
    multi-GT queries, queries without ground truth, a video without clips and NaN scores (same NaN policy)."""
    from dldkd_amd import eval as ev
    from dldkd_amd import scoring
    g = torch.Generator(device=DEV).manual_seed(nq + nv)
    lens = torch.randint(1, L + 1, (nv,), generator=g, device=DEV)
    if nv > 5:
        lens[5] = 0
    mask = (torch.arange(L, device=DEV)[None] < lens[:, None]).float()
    gal = [torch.randn(nv, L, 384, generator=g, device=DEV) * mask[..., None] for _ in range(nb)]
    qs = [torch.randn(nq, 384, generator=g, device=DEV) for _ in range(nb)]
    if nq > 20:
        qs[0][7] = float("nan")                               # a diverged query: NaN scores everywhere
    pg, pq = scoring.pack_gallery(gal, mask), scoring.pack_queries(qs)
    rs = np.random.RandomState(3)
    gts = {}
    for q in range(nq):
        k = rs.randint(0, 4) if q % 11 == 0 else 1             # some queries without / with several ground-truth videos
        if k:
            gts[q] = [int(v) for v in rs.choice(nv, size=min(k, nv), replace=False)]
    ptr, idx = ev.gt_csr(gts, nq, DEV)
    ws = scoring.simpool_partials(pq, pg)
    got = scoring.rank_partials(ws, pq, pg, ptr, idx).cpu()
    fused, s0, s1 = scoring.simpool_finish(ws, pq, pg, want_branches=True)
    mats = [s0, s1 if nb == 2 else s0, fused if nb == 2 else s0]
    for k, m in enumerate(mats):
        rb, rf = ev.gt_ranks_gpu(m, gts, (ptr, idx))
        assert torch.equal(got[k, 0], rb.cpu()), k
        assert torch.equal(got[k, 1], rf.cpu()), k
    if nq > 20:
        assert int(got[2, 0, 7]) == nv + 1                      # the NaN query ranks last


def test_eval_epoch_equals_matrix_path(golden_dir):
    """eval_epoch (ranks from the partial planes) returns exactly the SumR of the explicit matrix path of the reference's
    eval_epoch (compute_query2ctx_info -> fuse -> cal_perf, eval.py:243-263)."""
    from dldkd_amd import eval as ev
    m = _model(3072, 768, synth.make_params(51, 3072, 768))
    vids, txts = synth.make_eval_sets(5, nv=64, caps=3, dv=3072, dq=768)
    opt = _opt()
    with torch.no_grad():
        sumr = ev.eval_epoch(m, synth.ListDataset(list(vids)), synth.ListDataset(list(txts)), opt)
        ctx = ev.compute_context_info(m, synth.ListDataset(list(vids)), opt, keep_frame_feats=False)
        fused, s0, s1, metas = ev.score_queries(m, synth.ListDataset(list(txts)), opt, ctx)
    _, t2v = ev.get_gt(ctx["video_metas"], metas)
    r = ev.cal_perf(-fused, t2v)
    assert sumr == pytest.approx(r[0] + r[1] + r[2] + r[3], abs=1e-9)


@pytest.mark.parametrize("nq,nv,L,cuts", [(3000, 615, 128, (0, 200, 615)), (333, 77, 40, (0, 30, 30, 77)), (64, 9, 9, (0, 4, 9))])
def test_sharded_ranks_from_partial_planes_equal_unsharded(nq, nv, L, cuts):
    """The two local halves of eval_epoch_sharded's ranking (scoring.shard_thresholds / shard_counts: thresholds and counts straight
    from each shard's partial planes) combined the way dist.sharded_ranks_from_partials combines them (MAX, then SUM; here in
    one process, shards = separate packed galleries incl. an EMPTY shard) against scoring.rank_partials of the whole gallery:
    identical ranks for all three score kinds and both flavours, with multi-GT queries spread over shards, queries without
    ground truth and a NaN query (the NaN first-GT flag is exercised by tests/test_dist_cpu.py)."""
    from dldkd_amd import dist as ddist, eval as ev, scoring
    g = torch.Generator(device=DEV).manual_seed(nq + nv)
    lens = torch.randint(1, L + 1, (nv,), generator=g, device=DEV)
    mask = (torch.arange(L, device=DEV)[None] < lens[:, None]).float()
    gal = [torch.randn(nv, L, 384, generator=g, device=DEV) * mask[..., None] for _ in range(2)]
    qs = [torch.randn(nq, 384, generator=g, device=DEV) for _ in range(2)]
    qs[0][7] = float("nan")
    rs = np.random.RandomState(5)
    gts = {}
    for q in range(nq):
        k = rs.randint(0, 4) if q % 7 == 0 else 1
        if k:
            gts[q] = [int(v) for v in rs.choice(nv, size=min(k, nv), replace=False)]
    pq = scoring.pack_queries(qs)
    pg = scoring.pack_gallery(gal, mask)
    ptr, idx = ev.gt_csr(gts, nq, DEV)
    want = scoring.rank_partials(scoring.simpool_partials(pq, pg), pq, pg, ptr, idx).cpu().long()
    shards = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        pgs = scoring.pack_gallery([x[lo:hi] for x in gal], mask[lo:hi]) if hi > lo else scoring.GalleryPacker(0, L, 2, torch.device(DEV)).finish()
        ws = scoring.simpool_partials(pq, pgs)
        p_, i_, f_, has = ddist.local_gt_csr(gts, nq, lo, hi)
        shards.append((pgs, ws, [torch.from_numpy(a).to(DEV) for a in (p_, i_, f_)], has))
    parts = [scoring.shard_thresholds(ws, pq, pgs, *csr) for pgs, ws, csr, _ in shards]
    thr = torch.stack([p[0] for p in parts]).amax(0)             # all-reduce(MAX)
    flag = torch.stack([p[1] for p in parts]).amax(0)
    counts = sum(scoring.shard_counts(ws, pq, pgs, thr).long() for pgs, ws, _, _ in shards)      # all-reduce(SUM)
    ranks = torch.clamp(counts + 1, max=nv + 1)
    worst = (flag > 0) | (~torch.from_numpy(shards[0][3]).to(DEV) | (pq.bad[:nq] > 0))[None, None, :]
    ranks = torch.where(worst, torch.full_like(ranks, nv + 1), ranks).cpu()
    assert torch.equal(ranks, want)
    assert int(want[2, 0, 7]) == nv + 1                          # the NaN query ranks last on both paths


def test_eval_feature_cache_replays_without_the_loader():
    """train() evaluates the same validation sets after every epoch: the raw features of the first pass stay on the device and
    later passes never touch the datasets (eval.py _cached_batches).  Same SumR before and after a weight change as with the
    cache off; the second epoch reads no item; another dataset object, a changed length, or the cap miss the cache."""
    from dldkd_amd import eval as ev

    class Counting(synth.ListDataset):
        reads = 0

        def __getitem__(self, i):
            type(self).reads += 1
            return super().__getitem__(i)

    ev.clear_feature_cache()
    m = _model(3072, 768, synth.make_params(53, 3072, 768))
    vids, txts = synth.make_eval_sets(7, nv=40, caps=2, dv=3072, dq=768)
    dv, dt = Counting(list(vids)), Counting(list(txts))
    opt = _opt()
    opt.eval_precision = "throughput"
    off = types.SimpleNamespace(**vars(opt), eval_feature_cache=False)
    with torch.no_grad():
        ref1 = ev.eval_epoch(m, dv, dt, off)
        Counting.reads = 0
        s1 = ev.eval_epoch(m, dv, dt, opt)                  # fills the cache
        assert Counting.reads == 40 + 80
        s2 = ev.eval_epoch(m, dv, dt, opt)                  # replays it
        assert Counting.reads == 40 + 80 and s1 == s2 == ref1
        for p in m.parameters():                             # "an epoch of training"
            p.data.mul_(1.01)
        s3 = ev.eval_epoch(m, dv, dt, opt)
        assert Counting.reads == 40 + 80 and s3 == ev.eval_epoch(m, dv, dt, off)
        n = Counting.reads
        ev.eval_epoch(m, Counting(list(vids)), dt, opt)      # another gallery object: its own entry
        assert Counting.reads == n + 40
        tiny = types.SimpleNamespace(**vars(opt), eval_feature_cache_gb=1e-6)
        ev.clear_feature_cache()
        ev.eval_epoch(m, dv, dt, tiny)
        n = Counting.reads
        ev.eval_epoch(m, dv, dt, tiny)                        # nothing fitted under the cap: read again
        assert Counting.reads == n + 120
    ev.clear_feature_cache()


def test_resident_gallery_features_match_the_padded_super_batches(monkeypatch):
    """Throughput-mode eval_epoch keeps the gallery's raw features as a ragged bf16 table with the rows' LayerNorm statistics
    (eval.ResidentGallery / ops.ResidentRows) and encodes it with K4b + the fused tower kernel over the whole table.  Against
    the padded fp32 super-batches through K4 + the fused tower kernel (RESIDENT_FEATURES off): same lengths and visiting order,
    packed gallery rows within one bf16 step, same R@K; chunked tables (several K4b / K5 launch pairs) and the streaming form
    (table emptied every few clips: no cache) give the same; the kept table is 6 KB per clip."""
    from dldkd_amd import eval as ev, scoring
    m = _model(3072, 768, synth.make_params(61, 3072, 768))
    vids, txts = synth.make_eval_sets(11, nv=70, caps=2, dv=3072, dq=768)
    dv, dt = synth.ListDataset(list(vids)), synth.ListDataset(list(txts))
    opt = _opt()
    opt.eval_precision = "throughput"

    def run(**patch):
        ev.clear_feature_cache()
        for k, v in patch.items():
            monkeypatch.setattr(ev, k, v)
        with torch.no_grad(), ev.eval_precision(m, opt):
            info = ev.compute_context_info(m, dv, opt, keep_frame_feats=False)
            ranks, _ = ev.rank_queries(m, dt, opt, info)
        for k in patch:
            monkeypatch.undo()
        return info, ranks

    old, r_old = run(RESIDENT_FEATURES=False)
    new, r_new = run()
    n_clips = int(new["_packed"].lens.sum())
    res = next(iter(ev._FEATURE_CACHE[dv].values()))
    assert res.complete and res.table.rows == n_clips and res.table.nbytes() == n_clips * (3072 * 2 + 8)
    chunked, r_ch = run(RESIDENT_CHUNK_ROWS=100)
    assert len(next(iter(ev._FEATURE_CACHE[dv].values())).chunks) > 3
    off = types.SimpleNamespace(**vars(opt), eval_feature_cache=False)
    ev.clear_feature_cache()
    monkeypatch.setattr(ev, "RESIDENT_STREAM_ROWS", 150)
    with torch.no_grad(), ev.eval_precision(m, off):
        streamed = ev.compute_context_info(m, dv, off, keep_frame_feats=False)
        r_st, _ = ev.rank_queries(m, dt, off, streamed)
    assert dv not in ev._FEATURE_CACHE or not ev._FEATURE_CACHE[dv]
    for info, ranks in ((new, r_new), (chunked, r_ch), (streamed, r_st)):
        # (the padded super-batches round the mask's width up to whole 32-clip groups; the table's mask has the reference's width:
        # the longest video, eval.py:139-155)
        assert info["video_metas"] == old["video_metas"] and torch.equal(info["video_mask"].sum(1), old["video_mask"].sum(1))
        assert info["video_mask"].shape[1] == int(info["_packed"].lens.max())
        a, b = info["_packed"], old["_packed"]
        assert torch.equal(a.lens, b.lens) and torch.equal(a.order, b.order)
        for x, y in zip(a.blobs, b.blobs):
            xf, yf = x.view(torch.bfloat16).float(), y.view(torch.bfloat16).float()
            # (the table's path hands h0 to the tower as bf16 rows, the padded path as fp32: one more rounding, still within a bf16
            # step of a unit row, and a rank moves by at most one place)
            assert (xf - yf).abs().max().item() <= 2 ** -8 and (xf != yf).float().mean().item() < 0.15
        # (an untrained model: 70 near-tied scores per query, so a bf16 step on h0 swaps neighbours; what the rounding does to R@K
        # of a trained model is gated in test_rk_gate_gpu / profiles/r04/rk_gate.json, mode "resident")
        assert np.abs(ranks.astype(np.int64) - r_old.astype(np.int64)).max() <= 3
        assert np.abs(ranks.astype(np.int64) - r_old.astype(np.int64)).mean() < 0.5
    ev.clear_feature_cache()


def test_reused_gallery_buffers_equal_a_fresh_encode():
    """From the second epoch on, eval_epoch re-encodes the gallery IN PLACE into the buffers of the first cached epoch
    (eval.ResidentGallery.gallery_blobs: zero-filled once; the fused tower kernel then skips the rows past a video's last 16-row
    tile, ops.SKIP_ZERO_ROWS).  Bit for bit the packed gallery a fresh encode writes (every row, the padding included) - also
    after the model's weights changed between the epochs (nothing of the older encode survives in the valid rows) - and the R@K
    of the reference's per-batch path (method/eval.py:139-155) comes out the same."""
    from dldkd_amd import eval as ev
    vids, txts = synth.make_eval_sets(13, nv=90, caps=2, dv=3072, dq=768)
    dv, dt = synth.ListDataset(list(vids)), synth.ListDataset(list(txts))
    opt = _opt()
    opt.eval_precision = "throughput"
    models = [_model(3072, 768, synth.make_params(s, 3072, 768)) for s in (71, 72)]

    def epoch(m):
        with torch.no_grad(), ev.eval_precision(m, opt):
            info = ev.compute_context_info(m, dv, opt, keep_frame_feats=False)
            ranks, _ = ev.rank_queries(m, dt, opt, info)
        return [b.clone() for b in info["_packed"].blobs], ranks, info["_packed"]

    fresh = []
    for m in models:                                        # first epochs: buffers the kernel writes whole, padding included
        ev.clear_feature_cache()
        fresh.append(epoch(m))
    ev.clear_feature_cache()
    epoch(models[0])                                        # builds the resident table
    first = epoch(models[0])                                # cached: zero-filled buffers, padding rows skipped
    res = next(iter(ev._FEATURE_CACHE[dv].values()))
    kept = [b.data_ptr() for b in res.gallery_blobs]
    again = epoch(models[0])                                # in place
    other = epoch(models[1])                                # in place, other weights
    assert [b.data_ptr() for b in other[2].blobs] == kept
    for got, want in ((first, fresh[0]), (again, fresh[0]), (other, fresh[1])):
        assert all(torch.equal(x, y) for x, y in zip(got[0], want[0]))
        assert np.array_equal(got[1], want[1])
    assert any(not torch.equal(x, y) for x, y in zip(fresh[0][0], fresh[1][0]))
    ev.clear_feature_cache()


def test_ragged_first_pass_ingest_builds_the_same_table(monkeypatch):
    """The first pass over a gallery whose features start in host memory: items -> pinned staging ring -> table rows
    (eval.RESIDENT_RAGGED_INGEST, ops.ResidentRows.append_rows: no padded host batch) against the padded-batch form (collate_frame_val
    -> .to(device) -> ResidentRows.append): the same fp16 rows, statistics, lengths and ids bit for bit - with a staging ring smaller
    than a video (rows of one item arrive in several uploads) and with the streaming form (table emptied every few clips)."""
    from dldkd_amd import eval as ev, data
    m = _model(3072, 768, synth.make_params(63, 3072, 768))
    vids, _ = synth.make_eval_sets(17, nv=57, caps=1, dv=3072, dq=768)
    dv = synth.ListDataset(list(vids))
    opt = _opt()
    opt.eval_precision = "throughput"
    real = data._PinnedAppender

    def build(ragged, ring_bytes=None):
        ev.clear_feature_cache()
        monkeypatch.setattr(ev, "RESIDENT_RAGGED_INGEST", ragged)
        if ring_bytes:
            monkeypatch.setattr(data, "_PinnedAppender", lambda dev, chunks: real(dev, chunks, ring_bytes=ring_bytes))
        with torch.no_grad(), ev.eval_precision(m, opt):
            info = ev.compute_context_info(m, dv, opt, keep_frame_feats=False)
        monkeypatch.undo()
        res = next(iter(ev._FEATURE_CACHE[dv].values()))
        t = res.table
        return (t.xb[:t.rows].clone(), t.mean[:t.rows].clone(), t.rstd[:t.rows].clone(), list(t.lens), list(res.metas),
                [b.clone() for b in info["_packed"].blobs])

    ref = build(False)
    for ring in (None, 5 * 3072 * 4, "workers"):            # the default 64-MiB ring / a ring of five clip rows / loader workers
        if ring == "workers":                               # (a worker hands its batch over as ONE concatenated tensor)
            opt.num_workers, ring = 2, None
        got = build(True, ring)
        assert got[3] == ref[3] and got[4] == ref[4] and sum(got[3]) == got[0].shape[0]
        assert all(torch.equal(a, b) for a, b in zip(got[:3], ref[:3]))
        assert all(torch.equal(a, b) for a, b in zip(got[5], ref[5]))
    opt.num_workers = 0
    # streaming (no cache): the table is a staging buffer encoded and emptied every 150 clips, at the same loader-batch boundaries in
    # both forms - the packed gallery comes out the same
    off = types.SimpleNamespace(**vars(opt), eval_feature_cache=False)
    streamed = []
    for ragged in (False, True):
        ev.clear_feature_cache()
        monkeypatch.setattr(ev, "RESIDENT_RAGGED_INGEST", ragged)
        monkeypatch.setattr(ev, "RESIDENT_STREAM_ROWS", 150)
        with torch.no_grad(), ev.eval_precision(m, off):
            info = ev.compute_context_info(m, dv, off, keep_frame_feats=False)
        monkeypatch.undo()
        streamed.append(([b.clone() for b in info["_packed"].blobs], info["_packed"].lens.clone(), list(info["video_metas"])))
    assert all(torch.equal(a, b) for a, b in zip(streamed[0][0], streamed[1][0]))
    assert torch.equal(streamed[0][1], streamed[1][1]) and streamed[0][2] == streamed[1][2] == ref[4]
    ev.clear_feature_cache()
