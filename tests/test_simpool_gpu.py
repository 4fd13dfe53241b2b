"""GPU parity: HIP simpool (bf16 MFMA) against the oracle and the committed golden vectors.

Two yardsticks per case:
  * "exact-input": the oracle in fp64 on the SAME bf16-rounded normalised operands the kernel sees
    -> only fp32 accumulation order (and an occasional 1-ulp bf16 flip from the normalisation being
    summed in a different order on CPU and GPU) differs -> tolerance 1e-4.  Catches layout / mask / max bugs.
  * "reference": the oracle in fp32 on the original fp32 inputs (what the reference computes)
    -> adds bf16 operand rounding -> tolerance 1.5e-3 absolute on cosine scores in [-1, 1] (measured 4-5e-4; 6e-3 until round 5).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import dldkd_oracle as orc
import synth

pytestmark = pytest.mark.gpu

TOL_EXACT = 1e-4
TOL_BF16 = 1.5e-3


def _bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float64)


def _run(q_list, g_list, mask, normalize=True, w=(0.7, 0.3)):
    from dldkd_amd import scoring
    dev = "cuda:0"
    pq = scoring.pack_queries([q.to(dev) for q in q_list], normalize=normalize)
    pg = scoring.pack_gallery([g.to(dev) for g in g_list], None if mask is None else mask.to(dev), normalize=normalize)
    fused, s0, s1 = scoring.simpool_eval(pq, pg, w=w, want_fused=True, want_branches=True)
    torch.cuda.synchronize()
    return fused.cpu(), s0.cpu(), None if s1 is None else s1.cpu(), pg


def _oracle_exact(q, g, mask, normalize):
    if normalize:
        q = F.normalize(q.float(), dim=-1)
        g = F.normalize(g.float(), dim=-1)
    return orc.unnormalized_sim_scores(_bf16_round(q), _bf16_round(g), None if mask is None else mask.double())


def _check(q_list, g_list, mask, normalize=True, tol_ref=TOL_BF16):
    fused, s0, s1, pg = _run(q_list, g_list, mask, normalize)
    outs = [s0, s1]
    for b, (q, g) in enumerate(zip(q_list, g_list)):
        exact = _oracle_exact(q, g, mask, normalize)
        assert (outs[b].double() - exact).abs().max().item() <= TOL_EXACT * max(1.0, exact.abs().max().item()), f"branch {b} exact-input"
        ref = orc.sim_scores(q, g, mask)[0] if normalize else orc.unnormalized_sim_scores(q, g, mask)
        # raw scores: the cosine tolerance times the largest |q| |g| (a raw score IS its cosine times that product)
        scale = 1.0 if normalize else float(q.float().norm(dim=-1).max() * g.float().norm(dim=-1).max())
        assert (outs[b] - ref).abs().max().item() <= tol_ref * scale, f"branch {b} vs fp32 reference"
    if len(q_list) == 2:
        assert (fused - (0.7 * s0 + 0.3 * s1)).abs().max().item() <= 1e-6
    else:
        assert (fused - s0).abs().max().item() == 0
    if mask is not None:
        assert (pg.lens.cpu() == (mask > 0).sum(1).int()).all()
    return fused


def test_g1_golden(golden_dir):
    g = np.load(f"{golden_dir}/g1_simpool.npz")
    rs = np.random.RandomState(11)
    q = torch.from_numpy(rs.standard_normal((7, 384)).astype(np.float32))
    ctx = torch.from_numpy(rs.standard_normal((5, 9, 384)).astype(np.float32))
    mask = torch.from_numpy((np.arange(9)[None] < g["lens"][:, None]).astype(np.float32))
    ctx = ctx * mask.unsqueeze(-1)
    _, s0, _, _ = _run([q], [ctx], mask, True)
    assert np.abs(s0.numpy() - g["pooled"]).max() <= TOL_BF16
    _, r0, _, _ = _run([q], [ctx], mask, False)
    assert np.abs(r0.numpy() - g["raw"]).max() <= TOL_BF16 * float(q.norm(dim=-1).max() * ctx.norm(dim=-1).max())     # (cosine tolerance x |q| |g|)
    _, n0, _, _ = _run([q], [ctx], None, True)
    assert np.abs(n0.numpy() - g["pooled_nomask"]).max() <= TOL_BF16


@pytest.mark.parametrize("nq,nv,L,len_lo", [(64, 64, 16, 4), (1, 1, 1, 1), (33, 5, 128, 1), (200, 131, 128, 24),
                                            (97, 66, 40, 33), (31, 7, 96, 65), (5, 4, 128, 128)])
def test_vs_oracle_two_branches(nq, nv, L, len_lo):
    d0 = synth.make_gallery(100 + nq, nq, nv, L, len_lo, sigma=0.5)
    d1 = synth.make_gallery(200 + nq, nq, nv, L, len_lo, sigma=1.0)
    _check([d0["q"], d1["q"]], [d0["g"], d1["g"] * d0["mask"].unsqueeze(-1)], d0["mask"])


def test_single_branch_and_raw_mode():
    d = synth.make_gallery(7, 50, 37, 64, 3, sigma=0.3)
    _check([d["q"]], [d["g"]], d["mask"], normalize=True)
    _check([d["q"]], [d["g"]], d["mask"], normalize=False)


def test_all_negative_scores_not_beaten_by_padding():
    """Padded clips give dot = 0; a video whose valid clips all score < 0 must keep its negative max
    (the reference masks padding to -1e10, model.py:444-445)."""
    d = synth.make_gallery(9, 40, 9, 64, 5, sigma=0.0)
    g, mask = d["g"].clone(), d["mask"].clone()
    for v, n in ((2, 1), (5, 33), (7, 31)):          # every valid clip of these videos is anti-aligned with query v
        mask[v] = (torch.arange(64) < n).float()
        g[v] = (-d["q"][v].unsqueeze(0) + 0.05 * torch.randn(64, 384)) * mask[v].unsqueeze(-1)
    fused = _check([d["q"]], [g], mask)
    for v in (2, 5, 7):
        assert fused[v, v] < -0.9


def test_mask_none_equals_full_mask():
    d = synth.make_gallery(3, 20, 6, 32, 32, sigma=0.2)
    a = _run([d["q"]], [d["g"]], None)[1]
    b = _run([d["q"]], [d["g"]], torch.ones(6, 32))[1]
    assert torch.equal(a, b)


def test_empty_inputs():
    from dldkd_amd import scoring
    dev = "cuda:0"
    pq = scoring.pack_queries([torch.zeros(0, 384, device=dev)])
    pg = scoring.pack_gallery([torch.randn(3, 8, 384, device=dev)])
    fused, _, _ = scoring.simpool_eval(pq, pg)
    assert fused.shape == (0, 3)


def test_rank_parity_planted_gallery():
    """R@1/5/10/100 of HIP scores vs fp32-oracle scores on a planted-signal gallery agree within 0.1
    (BASELINE.json north_star gate), at a size the oracle finishes in seconds."""
    nq, nv = 2000, 1500
    d0 = synth.make_gallery(42, nq, nv, 128, 24, sigma=5.0)
    d1 = synth.make_gallery(43, nq, nv, 128, 24, sigma=6.0)
    g1 = d1["g"] * d0["mask"].unsqueeze(-1)
    fused, _, _, _ = _run([d0["q"], d1["q"]], [d0["g"], g1], d0["mask"])
    oi, oe = orc.eval_scores(d0["q"], d1["q"], d0["g"], g1, d0["mask"], chunk=50)
    of = orc.fuse_scores(oi, oe)
    gt = d0["gt"]

    def recalls(s):
        gts = s[torch.arange(nq), gt]
        rank = 1 + (s > gts.unsqueeze(1)).sum(1)
        return [100.0 * (rank <= k).float().mean().item() for k in (1, 5, 10, 100)]
    rh, ro = recalls(fused), recalls(of)
    assert 5.0 < ro[0] < 95.0, f"planted signal should give a non-trivial R@1, got {ro}"
    for a, b in zip(rh, ro):
        assert abs(a - b) <= 0.1, (rh, ro)


@pytest.mark.parametrize("nq,nv,L,len_lo", [(200, 131, 128, 24), (97, 300, 40, 1), (33, 64, 128, 128), (50, 1, 7, 7), (1000, 9, 64, 0)])
def test_query_split_is_bit_invariant(nq, nv, L, len_lo):
    """The launch grid is [query range][branch][4 videos]; any split gives the same matrix bit for bit (each score is the
    max of the same bf16 dot products accumulated in the same k order), including splits whose last range is short, and
    the per-range arrival counters end at (workgroups per range)."""
    from dldkd_amd import scoring
    d0 = synth.make_gallery(300 + nq, nq, nv, L, len_lo, sigma=0.5)
    d1 = synth.make_gallery(400 + nq, nq, nv, L, len_lo, sigma=1.0)
    dev = "cuda:0"
    pq = scoring.pack_queries([d0["q"].to(dev), d1["q"].to(dev)])
    pg = scoring.pack_gallery([d0["g"].to(dev), (d1["g"] * d0["mask"].unsqueeze(-1)).to(dev)], d0["mask"].to(dev))
    ref = scoring.simpool_finish(scoring.simpool_partials(pq, pg, q_split=1), pq, pg, want_branches=True)
    n_tiles = (nq + 31) // 32
    for split in (0, 2, 3, 5, n_tiles, n_tiles + 7):
        done = torch.zeros(64, dtype=torch.int32, device=dev)
        ws = scoring.simpool_partials(pq, pg, q_split=split, done=done if split else None)
        got = scoring.simpool_finish(ws, pq, pg, want_branches=True)
        for a, b in zip(got, ref):
            assert torch.equal(a, b), split
        if split:
            eff = min(split, n_tiles)
            tiles = (n_tiles + eff - 1) // eff
            n_ranges = (n_tiles + tiles - 1) // tiles
            want = torch.zeros(64, dtype=torch.int32)
            want[:n_ranges] = (pg.scorer_waves() + 3) // 4 * 2         # (pair waves: fewer than one wave per video)
            assert torch.equal(done.cpu(), want), (split, done[:8].tolist())
        # per-range finish assembles the same matrix
        if split in (3, 5):
            per = ((n_tiles + min(split, n_tiles) - 1) // min(split, n_tiles)) * 32
            rows = [scoring.simpool_finish(ws, pq, pg, q_range=(lo, min(lo + per, nq)))[0] for lo in range(0, nq, per)]
            assert torch.equal(torch.cat(rows, 0), ref[0])


@pytest.mark.parametrize("nq,nv,L,len_lo", [(100, 37, 128, 0), (257, 301, 128, 1), (64, 5, 32, 3), (1000, 2000, 128, 24), (33, 1, 128, 128),
                                            (500, 777, 100, 1), (129, 64, 64, 60), (70, 333, 128, 100), (40, 50, 128, 57)])
def test_pair_waves_are_bit_identical_to_one_video_per_wave(nq, nv, L, len_lo):
    """dldkd_simpool_eval_pairs_bf16 (two videos per wave where they fill its 128 rows: the default on ragged galleries) against
    dldkd_simpool_eval_bf16 on the same packed operands: the partial planes must be equal BIT FOR BIT - the same MFMA accumulation
    per clip row, the same maxima over the same clips - for every boundary position (len A mod 16, mod 4), zero-length videos,
    galleries where nothing pairs, with and without a query split and its arrival counters."""
    from dldkd_amd import scoring
    dev = "cuda:0"
    g = torch.Generator().manual_seed(1000 + nv)
    lens = torch.randint(len_lo, L + 1, (nv,), generator=g)
    mask = (torch.arange(L)[None, :] < lens[:, None]).float().to(dev)
    gs = [torch.randn(nv, L, 384, generator=g).to(dev) for _ in range(2)]
    pq = scoring.pack_queries([torch.randn(nq, 384, generator=g).to(dev) for _ in range(2)])
    pg = scoring.pack_gallery(gs, mask)
    n = 2 * nv * ((nq + 31) // 32 * 32)
    was = scoring.PAIR_WAVES
    try:
        scoring.PAIR_WAVES = False
        assert pg.scorer_waves() == nv
        ref = scoring.simpool_partials(pq, pg).view(torch.int32)[:n].clone()
        scoring.PAIR_WAVES = True
        plan, n_waves, n_paired = pg.pair_plan()
        for split in (0, 3):
            done = torch.zeros(8, dtype=torch.int32, device=dev) if split else None
            got = scoring.simpool_partials(pq, pg, q_split=split, done=done).view(torch.int32)[:n]
            assert torch.equal(got, ref), (split, n_waves, n_paired)
            if split:
                n_tiles = (nq + 31) // 32
                per = -(-n_tiles // min(split, n_tiles))
                n_ranges = -(-n_tiles // per)
                assert done.tolist() == [(pg.scorer_waves() + 3) // 4 * 2] * n_ranges + [0] * (8 - n_ranges)
    finally:
        scoring.PAIR_WAVES = was
    if len_lo in (1, 24, 57, 60):
        assert n_paired > 0 and n_waves == nv - n_paired


def test_planned_split_fills_the_chip():
    """Host cost model: the TVR gallery on one GPU needs no split; one rank's 615-video ActivityNet shard (308 workgroups
    on 256 CUs) is split so that the modelled rounds are >= 80 % full; min_split is honoured."""
    from dldkd_amd import scoring
    assert scoring.plan_query_split(10895, 21793, 2)[0] == 1
    n, per = scoring.plan_query_split(17505, 615, 2)
    wgs = 308 * n
    assert n >= 3 and per % 32 == 0 and n * per >= 17505 and wgs / (256 * -(-wgs // 256)) >= 0.8
    n8, per8 = scoring.plan_query_split(17505, 615, 2, min_split=8)
    assert n8 >= 8 and per8 % 32 == 0 and (n8 - 1) * per8 < 17505 <= n8 * per8


def test_video_without_valid_clips_scores_minus_1e10():
    """An all-zero mask row: the reference's mask_logits leaves -1e10 on every clip, so the max is -1e10 in both
    branches and in the fusion (0.7 + 0.3 of it)."""
    d = synth.make_gallery(11, 37, 9, 48, 5, sigma=0.3)
    mask = d["mask"].clone()
    mask[4] = 0.0
    g = d["g"] * mask.unsqueeze(-1)
    fused, s0, s1, _ = _run([d["q"], d["q"]], [g, g], mask, True)
    oi, _ = orc.eval_scores(d["q"], d["q"], g, g, mask)
    assert torch.equal(oi[:, 4], torch.full((37,), -1e10))
    assert torch.equal(s0[:, 4], oi[:, 4]) and torch.equal(s1[:, 4], oi[:, 4])
    assert (fused[:, 4] - (-1e10)).abs().max() <= 1024.0          # 0.7 x + 0.3 x in fp32: within an ulp of -1e10
    keep = [v for v in range(9) if v != 4]
    assert (s0[:, keep] - oi[:, keep]).abs().max() <= TOL_BF16


def test_fuzz_random_shapes_vs_oracle():
    """40 random problems (any Nq / Nv, L in 1..128, lens in 0..L including videos without clips, 1-2 branches,
    cosine or raw dot) against the oracle on the bf16-rounded operands: the exact-input bar of _check."""
    rs = np.random.RandomState(2024)
    for case in range(40):
        nq, nv, L = int(rs.randint(1, 260)), int(rs.randint(1, 150)), int(rs.randint(1, 129))
        nb, normalize = int(rs.randint(1, 3)), bool(rs.randint(0, 2))
        lens = rs.randint(0 if case % 5 == 0 else 1, L + 1, size=nv)
        mask = torch.from_numpy((np.arange(L)[None] < lens[:, None]).astype(np.float32))
        g = torch.Generator().manual_seed(case)
        qs = [torch.randn(nq, 384, generator=g) for _ in range(nb)]
        gs = [torch.randn(nv, L, 384, generator=g) * mask.unsqueeze(-1) for _ in range(nb)]
        fused, s0, s1, _ = _run(qs, gs, mask, normalize)
        for b, out in enumerate([s0, s1][:nb]):
            exact = _oracle_exact(qs[b], gs[b], mask, normalize)
            has = torch.from_numpy(lens > 0)
            scale = max(1.0, float(exact[:, has].abs().max())) if has.any() else 1.0
            assert (out.double()[:, has] - exact[:, has]).abs().max().item() <= TOL_EXACT * scale if has.any() else True, (case, b)
            assert bool((out[:, ~has] == -1e10).all()), (case, b)
        if nb == 1:
            assert torch.equal(fused, s0)


@pytest.mark.parametrize("nv", [1, 5, 64, 1023, 1024, 1025, 21793])
def test_visit_order_kernel_equals_stable_argsort(nv):
    """dldkd_order_by_len_desc (counting sort) = torch.argsort(lens, descending, stable): the scorer's visiting order."""
    from dldkd_amd import scoring
    g = torch.Generator().manual_seed(nv)
    lens = torch.randint(0, 129, (nv,), generator=g).to(torch.int32)
    if nv > 4:
        lens[:3] = torch.tensor([128, 0, 128])
    order, inv = scoring.visit_order(lens.to("cuda:0"))
    want = torch.argsort(lens, descending=True, stable=True).to(torch.int32)
    assert torch.equal(order.cpu(), want)
    assert torch.equal(inv.cpu()[want.long()], torch.arange(nv, dtype=torch.int32))


def test_planes_do_not_depend_on_when_the_scorer_runs():
    """K1 visits a range's query tiles in an order rotated by the chip-wide clock (K1_ROT: workgroups dispatched at different times
    stream the same tiles at the same time, which keeps them in L2).  Every tile's scores are stored where they belong, so launches
    at different times - different rotations in every workgroup - must give bit-identical planes, with and without pair waves and
    with a query split (the rotation stays inside each range)."""
    import time
    from dldkd_amd import scoring
    dev = "cuda:0"
    g = torch.Generator().manual_seed(77)
    nq, nv, L = 3000, 700, 128
    lens = torch.randint(1, L + 1, (nv,), generator=g)
    mask = (torch.arange(L)[None, :] < lens[:, None]).float().to(dev)
    gs = [torch.randn(nv, L, 384, generator=g).to(dev) for _ in range(2)]
    pq = scoring.pack_queries([torch.randn(nq, 384, generator=g).to(dev) for _ in range(2)])
    pg = scoring.pack_gallery(gs, mask)
    n = 2 * nv * ((nq + 31) // 32 * 32)
    was = scoring.PAIR_WAVES
    try:
        for pairs in (True, False):
            scoring.PAIR_WAVES = pairs
            for split in (0, 4):
                ref = None
                for rep in range(5):
                    done = torch.zeros(8, dtype=torch.int32, device=dev) if split else None
                    ws = torch.full((scoring.native.lib().dldkd_simpool_eval_workspace_bytes(nq, nv, 2),), 0x7f, dtype=torch.uint8, device=dev)
                    got = scoring.simpool_partials(pq, pg, ws, q_split=split, done=done).view(torch.int32)[:n].clone()
                    torch.cuda.synchronize()
                    time.sleep(0.0007 * (rep + 1))                     # a different clock phase for the next launch
                    if ref is None:
                        ref = got
                    assert torch.equal(got, ref), (pairs, split, rep)
    finally:
        scoring.PAIR_WAVES = was
