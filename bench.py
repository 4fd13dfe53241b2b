#!/usr/bin/env python3
"""bench.py - headline benchmark: query x video pairs scored / s on the TVR full gallery (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W          (any N: with N > 1 and no WORLD_SIZE in the environment it starts its
                                                            own ranks - a fresh `python -m torch.distributed.run` child - and
                                                            relays rank 0's JSON line and the child's exit code)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W     (the same ranks)

A "step" = one pass of the scoring hot path over the whole synthetic TVR-shaped workload (config C2,
SURVEY.md 8d): normalise + pack the 10,895 x 2 query vectors, score them against the resident bf16
gallery of 21,793 videos x <=128 clips x 384 dims x 2 branches (key-clip max-pool in registers), fuse the
branches 0.7/0.3 into the (Nq, Nv) fp32 score matrix.  The gallery is resident in HBM in its packed
bf16 form before the timed region (it is the output format of the gallery encoder).  N > 1: the gallery
is sharded by video across ranks, every rank scores all queries against its shard, and the step ends
with the RCCL all-gather of the score blocks, one query range at a time under the scoring of the next ranges (strong
scaling: total work fixed).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dl-dkd_amd"))

import torch  # noqa: E402

NQ, NV, L, LEN_LO, D, NB = 10895, 21793, 128, 24, 384, 2
SIGMA = (5.5, 6.5)        # planted-signal noise (per branch) -> TVR-like R@1
W_FUSE = (0.7, 0.3)       # eval.py:254
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
# Recalls of the synthetic C2 / C4 problems as ONE GPU measures them (python bench.py / tools/bench_c4_n1.py): the gallery and
# the queries are the same tensors for every rank count, so an N-GPU run must print exactly these from its assembled matrix.
RECALL_N1 = {"R@1": 13.144, "R@5": 22.726, "R@10": 27.48, "R@100": 48.196}
RECALL_C4_N1 = {"R@1": 16.715, "R@5": 28.495, "R@10": 34.299, "R@100": 58.395}


C2 = dict(name="C2", nq=NQ, nv=NV, L=L, len_lo=LEN_LO, seed=2, sigma=SIGMA,
          workload="C2: TVR full eval gallery text->video scoring (configs[1])")
C4 = dict(name="C4", nq=17505, nv=4917, L=128, len_lo=128, seed=4, sigma=SIGMA,
          workload="C4: ActivityNet full eval (i3d feats, 128-clip videos), gallery sharded across the ranks (configs[3])")
BLK = 64      # videos per seeded block: a video's clips depend on (config seed, block, branch) only - never on the rank count


def _lens_all(cfg):
    return torch.randint(cfg["len_lo"], cfg["L"] + 1, (cfg["nv"],), generator=torch.Generator().manual_seed(cfg["seed"]))


def synth_videos(dev, cfg, v_lo, v_hi):
    """Encoded clips of videos [v_lo, v_hi) for both branches + their lengths: the SAME tensors whichever rank (or shard
    layout) asks for them, so the N-GPU gallery is the 1-GPU gallery cut by video."""
    n = max(v_hi - v_lo, 0)
    gs = [torch.empty(n, cfg["L"], D, device=dev) for _ in range(NB)]
    for blk in range(v_lo // BLK, (v_hi + BLK - 1) // BLK if n else 0):
        lo, hi = max(blk * BLK, v_lo), min(blk * BLK + BLK, v_hi)
        for b in range(NB):
            gen = torch.Generator(device=dev).manual_seed(1_000_003 * cfg["seed"] + 2 * blk + b)
            gs[b][lo - v_lo:hi - v_lo] = torch.randn(BLK, cfg["L"], D, generator=gen, device=dev)[lo - blk * BLK:hi - blk * BLK]
    lens = _lens_all(cfg)[v_lo:v_hi].to(dev)
    return gs, lens


def synth_shard(dev, cfg, v_lo, v_hi, world=1, comm=None):
    """Gallery shard [v_lo, v_hi) + the queries, IDENTICAL on every rank: query q is planted on a valid clip of video q mod nv;
    the rank that holds that video contributes the clip, the others zeros, and one all-reduce(SUM) at setup gives every rank the
    same query set (each ground-truth video is local to exactly one rank)."""
    nq, nv, Lc = cfg["nq"], cfg["nv"], cfg["L"]
    gs, lens = synth_videos(dev, cfg, v_lo, v_hi)
    lens_all = _lens_all(cfg).to(dev)
    mask = (torch.arange(Lc, device=dev).unsqueeze(0) < lens.unsqueeze(1)).float()
    qgen = torch.Generator(device=dev).manual_seed(cfg["seed"] + 7)          # same stream on every rank
    gt = torch.arange(nq, device=dev) % nv
    lstar = (torch.rand(nq, generator=qgen, device=dev) * lens_all[gt].float()).long().clamp(max=Lc - 1)
    local = (gt >= v_lo) & (gt < v_hi)
    qs = []
    for b in range(NB):
        noise = torch.randn(nq, D, generator=qgen, device=dev)
        base = torch.zeros(nq, D, device=dev)
        base[local] = gs[b][(gt[local] - v_lo), lstar[local]]
        if world > 1:
            comm.all_reduce(base, "sum")
        qs.append(base + cfg["sigma"][b] * noise)
    return gs, mask, lens, qs, gt


def recalls(scores, gt):
    g = scores.gather(1, gt.unsqueeze(1))
    rank = 1 + (scores > g).sum(1)
    return [round(100.0 * (rank <= k).float().mean().item(), 3) for k in (1, 5, 10, 100)]


def cpu_baseline(gs, mask, qs, fused_gpu, nq_s=500, nv_s=8718):
    """The oracle (CPU restatement of eval.py:188-208 + :254, fp32, 50-query chunks) timed on this
    box's host cores on a bounded sample of the same workload; doubles as a parity check.

    Thread count: torch's default (= all logical cores) oversubscribes these 50-query chunks badly (measured on the
    256-core bench box: 25k pairs/s with 256 threads, 230k with 32), so the baseline first probes {all, 128, 64, 32, 16, 8}
    threads on a small slice and times the sample with the best - the CPU is not handicapped."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dldkd_oracle as orc
    all_cores = os.cpu_count() or 1
    g0, g1 = gs[0][:nv_s].cpu(), gs[1][:nv_s].cpu()
    m = mask[:nv_s].cpu()
    g0, g1 = g0 * m.unsqueeze(-1), g1 * m.unsqueeze(-1)
    q0, q1 = qs[0][:nq_s].cpu(), qs[1][:nq_s].cpu()
    probe = {}
    for thr in sorted({all_cores, 128, 64, 32, 16, 8}, reverse=True):
        if thr > all_cores:
            continue
        torch.set_num_threads(thr)
        orc.eval_scores(q0[:50], q1[:50], g0[:256], g1[:256], m[:256])      # warm-up
        t0 = time.perf_counter()
        orc.eval_scores(q0[:50], q1[:50], g0[:1500], g1[:1500], m[:1500], chunk=50)
        probe[thr] = 50 * 1500 / (time.perf_counter() - t0)
    best = max(probe, key=probe.get)
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    oi, oe = orc.eval_scores(q0, q1, g0, g1, m, chunk=50)
    ref = orc.fuse_scores(oi, oe)
    dt = time.perf_counter() - t0
    err = (fused_gpu[:nq_s, :nv_s].cpu() - ref).abs().max().item()
    out = dict(value=nq_s * nv_s / dt, unit="pairs/s", cores=best, kind="port",
               sample=f"{nq_s} queries x {nv_s} videos (first 2/5 of the C2 gallery), both branches + fusion, "
                      f"fp32, 50-query chunks like eval.py:188-208; {dt:.1f} s with {best} threads "
                      f"(best of the probe {({k: int(v) for k, v in probe.items()})} pairs/s; host has {all_cores} logical cores)")
    # back to a small host thread pool: with the probe's 256 OpenMP threads left alive, every blocking HIP wait of this process
    # (event / device synchronize) woke up 2-60 ms late on the GPU box, and the eval-stage extras measured the host, not the GPU
    torch.set_num_threads(min(all_cores, 8))
    try:
        out["c1"] = c1_cpu_vs_gpu(orc, fused_gpu.device)
    except Exception as e:   # noqa: BLE001
        out["c1"] = repr(e)
    return out, err


def c1_cpu_vs_gpu(orc, dev):
    """BASELINE configs[0]: 64 queries x 64 videos x 16 clips of raw i3d/CLIP-dim features through the WHOLE eval
    path (both towers + scoring + fusion), oracle on the host cores vs the HIP path, same weights and inputs."""
    import types
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import synth
    from dldkd_amd.model import DLDKD
    params = synth.make_params(41, 3072, 768)
    b = synth.make_train_batch(1, nv=64, caps=1, L=16, dv=3072, dq=768)
    v, vm, t, tm = b["student_videos"], b["student_videos_mask"], b["student_text"], b["student_text_mask"]

    def cpu():
        gi, ge = orc.encode_context(params, v, vm)
        qi, qe = orc.encode_query(params, t, tm)
        oi, oe = orc.eval_scores(qi, qe, gi, ge, vm)
        return orc.fuse_scores(oi, oe)
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(nthr, 8))      # 64-row tensors: more threads only add fork/join time (256 threads: 31 s)
    try:
        cpu()
        t0 = time.perf_counter()
        ref = cpu()
        cpu_s = time.perf_counter() - t0
        c1_threads = torch.get_num_threads()
    finally:
        torch.set_num_threads(nthr)
    cfg = types.SimpleNamespace(visual_input_size=3072, query_input_size=768, inheritance_hidden=384, exploration_hidden=384,
                                max_ctx_l=128, max_desc_l=30, input_drop=0.2, drop=0.2, n_heads=4, initializer_range=0.02,
                                margin=0.1, use_hard_negative=False, hard_pool_size=20, label_style="soft")
    opt_ = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                 collection="tvr", alpha=0.8, belta=0.8)
    m = DLDKD(cfg, opt_)
    m.load_state_dict(params, strict=True)
    m = m.to(dev).eval()
    dv, dvm, dt_, dtm = v.to(dev), vm.to(dev), t.to(dev), tm.to(dev)

    def gpu():
        with torch.no_grad():
            gi, ge = m.encode_context(dv, dvm)
            qi, qe = m.encode_query(dt_, dtm)
            return m.pooled_scores([qi, qe], [gi, ge], dvm, want_branches=False)[0]
    gpu()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fused = gpu()
    torch.cuda.synchronize()
    gpu_s = (time.perf_counter() - t0) / 10
    return {"config": "C1 (configs[0]): 64 q x 64 v x 16 clips, raw 3072/768-d features -> towers -> scores -> fusion",
            "cpu_oracle_s": cpu_s, "cpu_threads": c1_threads, "hip_s": gpu_s, "pairs_per_s_cpu": 4096 / cpu_s, "pairs_per_s_hip": 4096 / gpu_s,
            "max_abs_err": (fused.cpu() - ref).abs().max().item()}


def extras(dev):
    """Secondary measurements the survey asks to report beside the headline: C3 training step and the
    gallery-encode rate.  Never allowed to break the headline line: failures are reported as strings."""
    import types
    out = {}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import synth
        from dldkd_amd import ops
        from dldkd_amd.model import DLDKD
        cfg = types.SimpleNamespace(visual_input_size=3072, query_input_size=768, inheritance_hidden=384,
                                    exploration_hidden=384, max_ctx_l=128, max_desc_l=30, input_drop=0.2, drop=0.2, n_heads=4,
                                    initializer_range=0.02, margin=0.1, use_hard_negative=True, hard_pool_size=20,
                                    label_style="soft")
        opt_ = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04,
                                     explore_nce_weight=0.04, collection="tvr", alpha=0.8, belta=0.8)
        # C3 / C5 training step (tools/bench_train.py): >= 30 timed steps after >= 10 warm-ups per mode, HIP events around
        # every step, median and p90; eager (train.train_step) and hipGraph-replayed (train.GraphedTrainStep, what
        # train.train() runs).  c3_train_step_ms[_bf16] = the replayed step's median stream time incl. float(loss).
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_train
        for cfgname, key in (("c3", "c3"), ("c5", "c5")):
            for prec in ("fp32", "bf16", "mixed"):
                r = bench_train.run(cfgname, prec, 0.2 if cfgname == "c3" else 0.15, steps=30, warmup=10, dev=str(dev),
                                    modes=("graph",) if prec == "mixed" else ("eager", "graph"))
                out[f"{key}_train_step_{prec}"] = r
                out[f"{key}_train_step_ms" + ("" if prec == "fp32" else f"_{prec}") if key == "c3" else f"c5_train_step_ms_{prec}"] = \
                    r["graph"]["stream_ms_median"]
                # what train.train() runs: the loss stays on the device (one read-back per epoch), the host runs ahead of the GPU
                out[f"{key}_train_step_ms_{prec}_deferred_loss"] = r["graph_no_loss_sync"]["stream_ms_median"]
                # ... and the batch already sits in the stepper's input buffers (a data path that fills them itself: no staging copy)
                if "graph_static_inputs" in r:
                    out[f"{key}_train_step_ms_{prec}_deferred_loss_static_inputs"] = r["graph_static_inputs"]["stream_ms_median"]
        # the training LOOP (train.train_epoch on a device-resident training set of TVR shapes: data path, per-epoch schedule,
        # loss read-back included), wall-clock per step - what train.train() delivers, beside the bare step above
        try:
            import prof_train_epoch
            from bench_train_loader import SynthTrainSet
            ds_loop = SynthTrainSet(1024)
            for prec in ("bf16", "mixed"):
                r = prof_train_epoch.run(1024, prec, dev=str(dev), epochs=6, ds=ds_loop)
                w = sorted(r["ms_per_step_wall"])
                out[f"c3_train_loop_ms_per_step_{prec}"] = {"median_epoch": w[len(w) // 2], "epochs": r["ms_per_step_wall"], "steps_per_epoch": r["steps_per_epoch"],
                                                            "captures": r["captures"], "eager_steps": r["eager_steps"], "replays": r["replays"],
                                                            "prefetched": r["prefetched"], "fallbacks": r["fallbacks"]}
            del ds_loop
            # ... and on a Charades-shaped set (C5's shapes; 1..6 captions per video: every batch has its own query count - the
            # stepper pads the query axis to a bucket of 32 and the fused losses skip the padding rows, so a handful of captures serve
            # every batch; unpadded, most steps of such a set ran eagerly or re-captured)
            ds_loop = SynthTrainSet(2048, config="c5")
            for prec in ("bf16", "mixed"):
                r = prof_train_epoch.run(2048, prec, dev=str(dev), epochs=8, ds=ds_loop, config="c5")
                w = sorted(r["ms_per_step_wall"])
                out[f"c5_train_loop_ms_per_step_{prec}"] = {"median_epoch": w[len(w) // 2], "epochs": r["ms_per_step_wall"], "steps_per_epoch": r["steps_per_epoch"],
                                                            "distinct_query_counts": len(r["queries_per_batch"] or []), "captures": r["captures"],
                                                            "eager_steps": r["eager_steps"], "replays": r["replays"], "prefetched": r["prefetched"],
                                                            "fallbacks": r["fallbacks"]}
            del ds_loop
        except Exception as ex:   # noqa: BLE001 - an extra never takes the headline line down
            out["c3_train_loop_ms_per_step"] = {"error": repr(ex)[:300]}
        out["c3_train_step_config"] = "TVR: 128 videos / 640 queries, L<=128, label_style=soft, hard negatives, dropout 0.2, " \
                                      "zero_grad + forward + backward + fused BertAdam; fp32 = parity mode (fp32-grade GEMMs: three bf16 " \
                                      "planes per operand, losses within 1e-4 of the reference), bf16 = every GEMM on bf16 MFMA with fp32 " \
                                      "accumulation (fp32 master weights / activations, losses 2e-2), mixed = fp32-grade forward on two bf16 planes per operand " \
                                      "+ the fused bf16 backward (losses within 1e-6 of the reference's, gradients ~1e-2); *_ms = hipGraph-replayed step, median"
        out["c5_train_step_config"] = "Charades rank-local step: 128 videos / 257 queries, L<=64, Dv=Dq=1024, dropout 0.15 " \
                                      "(the gradient all-reduce of the 17.5 MB flat bucket is not part of a 1-GPU run)"
        torch.manual_seed(0)
        m = DLDKD(cfg, opt_).to(dev)
        m.eval()
        B, Lc = 200, 128
        feats = torch.nn.functional.normalize(torch.randn(B, Lc, 3072, device=dev), dim=-1)
        mask = torch.ones(B, Lc, device=dev)
        with torch.no_grad():
            m.encode_context(feats, mask)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                m.encode_context(feats, mask)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        out["gallery_encode_videos_per_s"] = B / dt
        out["gallery_encode_raw_feature_GBps"] = B * Lc * 3072 * 4 / dt / 1e9
        out["gallery_encode_config"] = "200 x 128 clips x 3072-d fp32 features, both branches, parity-mode towers (fp32-grade three-plane GEMMs, fp32 attention)"
        m.fast_input_proj = True          # K4: bf16 input projection, LayerNorm folded, one pass over the features
        with torch.no_grad():
            m.encode_context(feats, mask)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                m.encode_context(feats, mask)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        out["gallery_encode_videos_per_s_k4_bf16"] = B / dt
        ops.set_gemm_precision("bf16")    # + every remaining tower GEMM on bf16 MFMA
        try:
            with torch.no_grad():
                m.encode_context(feats, mask)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    m.encode_context(feats, mask)
                torch.cuda.synchronize()
            out["gallery_encode_videos_per_s_all_bf16"] = B / ((time.perf_counter() - t0) / 5)
            # the eval driver groups loader batches into 1024-video super-batches (eval.CONTEXT_SUPER_BATCH)
            fb = torch.nn.functional.normalize(torch.randn(1024, Lc, 3072, device=dev), dim=-1)
            mb = torch.ones(1024, Lc, device=dev)
            with torch.no_grad():
                m.encode_context(fb, mb)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    m.encode_context(fb, mb)
                torch.cuda.synchronize()
            out["gallery_encode_videos_per_s_all_bf16_superbatch1024"] = 1024 / ((time.perf_counter() - t0) / 3)
            del fb, mb
        finally:
            ops.set_gemm_precision("fp32")
        del m
        torch.cuda.empty_cache()
        # C4 (configs[3]) on ONE GPU: ActivityNet gallery 4917 videos x 128 clips (all valid) x 17505 queries
        from dldkd_amd import scoring
        g4 = torch.Generator(device=dev).manual_seed(4)
        gal4 = [torch.randn(4917, 128, 384, generator=g4, device=dev) for _ in range(2)]
        q4 = [torch.randn(17505, 384, generator=g4, device=dev) for _ in range(2)]
        pg4 = scoring.pack_gallery(gal4, None)
        del gal4
        for _ in range(2):
            scoring.simpool_eval(scoring.pack_queries(q4), pg4, want_branches=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            scoring.simpool_eval(scoring.pack_queries(q4), pg4, want_branches=False)
        torch.cuda.synchronize()
        dt4 = (time.perf_counter() - t0) / 10
        out["c4_activitynet_1gpu"] = {"ms_per_step": dt4 * 1e3, "pairs_per_s": 17505 * 4917 / dt4,
                                      "algorithmic_TFLOPs": 2.0 * 384 * 2 * 17505 * 4917 * 128 / dt4 / 1e12,
                                      "config": "4917 videos x 128 clips (all valid) x 17505 queries, 2 branches + fusion, 1 GPU"}
        del pg4, q4
        torch.cuda.empty_cache()
        # single-GPU proxy of the 8-GPU ActivityNet run: one rank's 615-video shard x all queries vs the whole gallery
        from bench_shard_proxy import shard_proxy
        out["c4_shard_proxy_1_of_8"] = shard_proxy(str(dev))
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from bench_eval_e2e import stage_times
        # GPU work of one eval_epoch at C2 from RAW features (encode gallery + queries, score, rank); SURVEY 8d
        out["eval_epoch_gpu_stages_fp32"] = stage_times(NV, NQ, "fp32", str(dev))
        out["eval_epoch_gpu_stages_fast"] = stage_times(NV, NQ, "fast", str(dev))
        # round 3: throughput-mode gallery encode = K4 (row groups: padding skipped) + the fused tower kernel K5 straight into the
        # packed bf16 gallery, ragged lengths U{24..128}
        out["gallery_encode_videos_per_s_fused_k4_k5"] = out["eval_epoch_gpu_stages_fast"].get("gallery_videos_per_sec")
        # round 3, second half: what eval_epoch runs now - the gallery's raw features resident as a ragged fp16 table with the
        # rows' LayerNorm statistics (filled once), K4b over the whole table + the fused tower kernel over all videos
        out["eval_epoch_gpu_stages_resident"] = stage_times(NV, NQ, "resident", str(dev))
        # round 6: the wall-clock of the metric's own top-level call, host side included - dldkd_amd.eval.eval_epoch on in-memory
        # datasets of C2 size (method/data_provider.py:307-309,344-354's protocol) - and the oracle's eval on a stated sample beside it
        try:
            import bench_eval_epoch_c2
            out["eval_epoch_wall_s_c2"] = bench_eval_epoch_c2.run(str(dev), NV, NQ)
        except Exception as ex:   # noqa: BLE001 - an extra never takes the headline line down
            out["eval_epoch_wall_s_c2"] = {"error": repr(ex)[:300]}
        out["gallery_encode_videos_per_s_resident_k4b_k5"] = out["eval_epoch_gpu_stages_resident"].get("gallery_videos_per_sec")
        import types as _t
        cfg2 = cfg
        torch.manual_seed(0)
        m = DLDKD(cfg2, opt_).to(dev).eval()
        # host -> device rate of raw features (the DataLoader side of compute_context_info, eval.py:130-134): what bounds
        # an eval_epoch whose features start in host memory
        hb = torch.empty(256, 128, 3072, dtype=torch.float32).pin_memory()          # 403 MB, one 256-video batch
        db = torch.empty_like(hb, device=dev)
        db.copy_(hb, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            db.copy_(hb, non_blocking=True)
        torch.cuda.synchronize()
        gbps = 5 * hb.numel() * 4 / (time.perf_counter() - t0) / 1e9
        out["h2d_pinned_GBps"] = gbps
        out["gallery_encode_videos_per_s_pcie_bound"] = gbps * 1e9 / (128 * 3072 * 4)
        del hb, db
        xk = torch.nn.functional.normalize(torch.randn(400000, 3072, device=dev), dim=-1)      # 4.9 GB: beyond the L3
        fold = ops.FoldedInProj([m.visual_input_proj, m.exp_visual_input_proj])
        def k4_ms():                                   # median of 10 single launches after 3 warm-ups (HIP events)
            for _ in range(3):
                ops.in_proj_h16(xk, fold)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
            evs[0].record()
            for i in range(10):
                ops.in_proj_h16(xk, fold)
                evs[i + 1].record()
            torch.cuda.synchronize()
            return sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(10))[5]
        ms = k4_ms()
        byts = 400000 * 3072 * 4 + 400000 * 768 * 4 + 768 * 3072 * 2
        out["k4_in_proj_roofline"] = {"bound": "hbm", "achieved": byts / ms / 1e6, "peak": 8000.0, "unit": "GB/s",
                                      "frac": byts / ms / 1e6 / 8000.0, "kernel": "in_proj_rows128_kernel (dldkd_in_proj_h16_rows128)", "kernel_ms": ms,
                                      "shape": "400000 rows x 3072 fp32 -> 2 x 384 fp32", "timing": "median of 10 launches after 3 warm-ups"}
        # K4b on the same rows in their resident form (fp16 + row statistics): 307 -> 560 flop per byte, the MFMA pipe is its bound
        tab = ops.ResidentRows(3072, dev, 400000)
        for lo in range(0, 400000, 50000):
            tab.append(xk[lo:lo + 50000].view(50, 1000, 3072), [1000] * 50)
        outs_b = [torch.empty(400000, 384, device=dev) for _ in range(2)]
        def k4b_ms():
            for _ in range(3):
                ops.in_proj_resident(tab, 0, 400000, fold, out=outs_b)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
            evs[0].record()
            for i in range(10):
                ops.in_proj_resident(tab, 0, 400000, fold, out=outs_b)
                evs[i + 1].record()
            torch.cuda.synchronize()
            return sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(10))[5]
        msb = k4b_ms()
        flops = 2.0 * 400000 * 3072 * 768
        bytes_b = 400000 * 3072 * 2 + 400000 * 8 + 400000 * 768 * 4 + 768 * 3072 * 2
        out["k4b_in_proj_roofline"] = {"bound": "mfma", "achieved": flops / msb / 1e9, "peak": 2500.0, "unit": "TFLOP/s",
                                       "frac": flops / msb / 1e9 / 2500.0, "kernel": "in_proj_rows128b_kernel (dldkd_in_proj_h16_rows128b)",
                                       "kernel_ms": msb, "shape": "400000 rows x 3072 fp16 (+ mean, rstd) -> 2 x 384 fp32",
                                       "algorithmic_GB": bytes_b / 1e9, "hbm_GBps": bytes_b / msb / 1e6,
                                       "same_rows_fp32_k4_ms": ms, "timing": "median of 10 launches after 3 warm-ups"}
        del tab, outs_b
        ops.INPROJ_KERNEL = "full"                 # the round-1 kernel on the same box, same input
        try:
            out["k4_in_proj_roofline"]["round1_kernel_ms"] = k4_ms()
        finally:
            ops.INPROJ_KERNEL = "rows128"
    except Exception as e:   # noqa: BLE001
        out["error"] = repr(e)
    return out


def mfma_sustained():
    """tools/micro/mfma_peak (built by __graft_entry__.build()): the bf16 MFMA rate this chip sustains with REGISTER
    operands only - no LDS, no memory - on all-zero and on random data, for the scorer's instruction.  On random data
    the clock under load, not the issue rate, sets the ceiling; it is reported beside the spec peak, never instead of it."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "micro", "mfma_peak")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=120)
        out = {}
        for line in r.stdout.splitlines():
            mm = re.match(r"(zero|random)\s+operands, mfma_(\S+):\s+[\d.]+ ms\s+(\d+) TFLOP/s", line)
            if mm:
                out[f"{mm.group(1)}_{mm.group(2)}"] = float(mm.group(3))
        return out or None
    except Exception:   # noqa: BLE001
        return None


def run_sharded(cfg, dev, rank, world, comm, steps, warmup):
    """One workload through dist.OverlappedShardScorer (what --gpus N runs): the gallery cut by video, ONE scorer launch per
    step whose query ranges complete in order, per-range finish on a side stream + RCCL all_gather (comm.RcclComm: one enqueue per
    range on the collectives' stream).  Runs on every rank;
    returns the measurements (rank 0 adds the self-checks: recalls of the ASSEMBLED matrix and a sampled recompute)."""
    from dldkd_amd import dist as ddist
    from dldkd_amd import scoring
    nq, nv = cfg["nq"], cfg["nv"]
    shard = (nv + world - 1) // world
    v_lo, v_hi = min(rank * shard, nv), min((rank + 1) * shard, nv)
    gs, mask, lens, qs, gt = synth_shard(dev, cfg, v_lo, v_hi, world, comm)
    n_loc = v_hi - v_lo
    if n_loc < shard:   # pad the last shard with 1-clip zero videos so all_gather blocks are equal
        pad = shard - n_loc
        gs = [torch.cat([g, torch.zeros(pad, cfg["L"], D, device=dev)]) for g in gs]
        mask = torch.cat([mask, torch.zeros(pad, cfg["L"], device=dev)])
        mask[n_loc:, 0] = 1.0
    t0 = time.perf_counter()
    pg = scoring.pack_gallery(gs, mask)            # resident bf16 gallery (outside the timed region)
    torch.cuda.synchronize()
    pack_ms = (time.perf_counter() - t0) * 1e3
    del gs
    # >= 4 ranges so that the all-gather of range r (RCCL stream, parked on the range's arrival counter) runs under the
    # scoring of ranges r+1..; the split is the kernel's own (HipShardBackend asks the library's planner)
    backend = ddist.HipShardBackend(qs, pg, min_ranges=4, w=W_FUSE)
    overlap = ddist.OverlappedShardScorer(backend, backend.bounds, shard, dev, comm=comm)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]

    def fence():
        comm.host_wait(what="bench step loop")      # deadline-bounded (comm.py): a dead peer must end the run, not hang it
        comm.barrier()
        torch.cuda.synchronize()
    for _ in range(warmup):
        overlap.step()
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i][0].record()
        overlap.step()
        ev[i][1].record()
    fence()
    dt = time.perf_counter() - t0
    dt = comm.max_over_ranks(dt, dev)
    res = {"ms_per_step": dt / steps * 1e3, "value": nq * nv * steps / dt, "step_stream_ms": sum(s_.elapsed_time(e) for s_, e in ev) / steps,
           "n_ranges": len(overlap.bounds), "shard_videos": shard, "gallery_pack_ms_untimed": round(pack_ms, 2),
           "flops_per_step_all_ranks": 2.0 * D * NB * nq * float(_lens_all(cfg).sum().item())}
    if rank == 0:
        full = overlap.assemble(nv)                                   # (nq, nv) fp32 as every rank holds it after the gathers
        res["recall_hip"] = dict(zip(("R@1", "R@5", "R@10", "R@100"), recalls(full, gt)))
        # sampled recompute: one BLK-video block out of every rank's shard (the last one ends at the last video), scored by a
        # plain one-launch scorer call on THIS rank from re-generated clips, against the same columns of the assembled matrix
        worst, n_s = 0.0, 0
        for r in range(world):
            lo = min(r * shard, nv)
            hi = min(lo + BLK, min((r + 1) * shard, nv))
            if r == world - 1:
                lo, hi = max(nv - BLK, min(r * shard, nv)), nv
            if hi <= lo:
                continue
            g_s, lens_s = synth_videos(dev, cfg, lo, hi)
            m_s = (torch.arange(cfg["L"], device=dev).unsqueeze(0) < lens_s.unsqueeze(1)).float()
            ref = scoring.simpool_eval(scoring.pack_queries(qs), scoring.pack_gallery(g_s, m_s), W_FUSE)[0]
            worst = max(worst, (full[:, lo:hi] - ref).abs().max().item())
            n_s += hi - lo
        res["assembled_max_abs_diff"] = worst
        res["assembled_check"] = f"{n_s} videos ({BLK} per shard) x all {nq} queries recomputed on rank 0 by a plain one-launch scorer call"
        del full
    del overlap, backend, pg
    torch.cuda.empty_cache()
    return res


def c5_rank_batch(rank, device="cpu"):
    """One rank's Charades-STA batch (BASELINE configs[4]; /root/reference/do_charades.sh:6-14): 128 videos, captions per video
    [3, 2, 2, ...] = 257 queries, 8..64 clips, 1024-d student features, seed 5 + rank (every rank draws its own batch)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import synth
    caps = sorted([3] + [2] * 127, reverse=True)
    b = synth.make_train_batch(5 + rank, nv=128, caps=caps, L=64, len_lo=8, dv=1024, dq=1024)
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in b.items()}


def replicas_max_abs_diff(flat, comm):
    """max over ranks and elements of |this rank's parameters - rank 0's|: 0.0 iff the replicas are identical (all-reduce MAX, so
    every rank returns the same number).  `comm`: a dldkd_amd.comm communicator (gloo group in the CPU tests, RCCL on the GPU)."""
    ref = flat.detach().clone()
    comm.broadcast(ref, src=0)
    d = (flat.detach() - ref).abs().max().reshape(1).to(torch.float64)
    comm.all_reduce(d, "max")
    return float(d.item())


def run_c5_ddp(dev, rank, world, comm, steps, warmup):
    """BASELINE configs[4]: the Charades-STA training step, data parallel over the ranks (method/train.py:147-151 under DDP): every
    rank steps on ITS batch, the flat gradient buffer is mean-all-reduced over RCCL, the fused BertAdam update follows - replayed by
    train.GraphedTrainStep exactly as train() runs it.  Measured per gradient layout (one bucket = the default; tower buckets =
    opt.ddp_bucketed_overlap): step ms (max over ranks), the same step with the collective taken out (the rank-local replay), their
    difference = the exposed all-reduce time; self-check: after the timed steps the replicas' parameters are bit-identical."""
    import types
    from dldkd_amd import ops
    from dldkd_amd import train as T
    from dldkd_amd.model import DLDKD
    cfg = types.SimpleNamespace(visual_input_size=1024, query_input_size=1024, inheritance_hidden=384, exploration_hidden=384,
                                max_ctx_l=128, max_desc_l=30, input_drop=0.15, drop=0.15, n_heads=4, initializer_range=0.02,
                                margin=0.2, use_hard_negative=True, hard_pool_size=20, label_style="soft")
    mopt = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                 collection="charades", alpha=0.8, belta=0.8)
    batch = c5_rank_batch(rank, dev)
    res = {"workload": "Charades-STA training step (128 videos / 257 queries per rank, 1024-d, dropout 0.15), DDP over RCCL, bf16 GEMMs",
           "n_gpus": world, "grad_bytes": None}

    def fence():
        comm.host_wait(what="bench step loop")      # deadline-bounded (comm.py): a dead peer must end the run, not hang it
        comm.barrier()
        torch.cuda.synchronize()

    def timed(stepper):
        for _ in range(warmup):
            stepper(batch)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            stepper(batch)
        fence()
        return comm.max_over_ranks(time.perf_counter() - t0, dev) / steps * 1e3

    ops.set_gemm_precision("bf16")
    old_min = T.DDP_MIN_WORLD
    try:
        for name, bucketed in (("single_bucket", False), ("tower_buckets", True)):
            topt = types.SimpleNamespace(grad_clip=-1, lr=2.4e-4, wd=0.01, lr_warmup_proportion=0.01, n_epoch=100,
                                         ddp_bucketed_overlap=bucketed, seed=0)
            torch.manual_seed(0)                            # same initial weights on every rank; rank 0's are broadcast anyway
            m = DLDKD(cfg, mopt).to(dev).train()
            T.DDP_MIN_WORLD = min(old_min, max(world, 1))   # the one-rank test hook drives the data-parallel branch too
            optim = T.make_optimizer(m, topt, 1000)
            from dldkd_amd import dist as ddist
            ddist.broadcast_parameters(optim.fp)
            T.seed_rank(topt, rank)                         # dropout masks / triplet negatives differ between the replicas
            ddp_ms = timed(T.GraphedTrainStep(m, optim, topt, defer_loss_float=True))
            diff = replicas_max_abs_diff(optim.fp.flat, comm)
            # the same replayed step without the collective (the data-parallel branch switched off): what the all-reduce costs on top
            T.DDP_MIN_WORLD = world + 1
            local_ms = timed(T.GraphedTrainStep(m, optim, topt, defer_loss_float=True))
            res[name] = {"step_ms": ddp_ms, "step_ms_without_allreduce": local_ms, "exposed_allreduce_ms": ddp_ms - local_ms,
                         "replicas_max_abs_param_diff_after_timed_steps": diff, "replicas_identical": diff == 0.0,
                         "n_buckets": len(optim.fp.bucket_ranges)}
            res["grad_bytes"] = int(optim.fp.grad.numel() * 4)
            del m, optim
            torch.cuda.empty_cache()
    finally:
        T.DDP_MIN_WORLD = old_min
        ops.set_gemm_precision("fp32")
    return res


def launch_command(n, argv, port, python=None):
    """The command `bench.py --gpus N` starts when nobody has started its ranks: one rank per GPU of this node under
    torch.distributed.run (rendezvous on 127.0.0.1 - the boxes' hostname need not resolve), each rank = this script, same flags."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n)}",
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def relay_child(cmd, timeout_s=3600.0, env=None, out=None, err=None):
    """Run `cmd` as a FRESH child process (never exec: this parent may not touch the GPU and must not be replaced), pass its stderr
    through, keep its stdout back, and when it ends print the child's stdout with its last JSON line LAST - rank 0's result line
    becomes this process's last stdout line whatever banners the ranks wrote around it.  Returns the child's exit code; a child
    still running after timeout_s is killed with its whole process group (the ranks) and the code is 124."""
    import signal
    import subprocess
    out, err = out or sys.stdout, err or sys.stderr
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None if err is sys.stderr else subprocess.PIPE, text=True, env=env,
                         start_new_session=True)
    try:
        so, se = p.communicate(timeout=timeout_s if timeout_s and timeout_s > 0 else None)
        rc = p.returncode
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            p.kill()
        so, se = p.communicate()
        rc = 124
        print(f"bench.py: the launched ranks did not finish within {timeout_s:g} s: killed", file=err, flush=True)
    if se:
        err.write(se)
    lines = (so or "").splitlines()
    last_json = None
    for i in range(len(lines) - 1, -1, -1):
        t = lines[i].strip()
        if t.startswith("{") and t.endswith("}"):
            try:
                json.loads(t)
            except ValueError:
                continue
            last_json = i
            break
    for i, ln in enumerate(lines):
        if i != last_json:
            print(ln, file=err)                     # whatever else the ranks printed: not part of the result
    if last_json is not None:
        print(lines[last_json], file=out, flush=True)
    elif rc == 0:
        print("bench.py: the launched ranks exited 0 without a JSON line", file=err, flush=True)
        rc = 1
    if rc < 0:                                      # died by a signal
        rc = 128 - rc
    return rc


def self_launch(a, argv):
    """`python bench.py --gpus N` (N > 1) with no WORLD_SIZE in the environment: count the GPUs WITHOUT initialising one
    (torch.cuda.device_count()), refuse within seconds if there are fewer than N, else start the ranks and relay their result."""
    have = torch.cuda.device_count()
    if have < a.gpus:
        print(f"bench.py: --gpus {a.gpus} needs {a.gpus} GPUs on this node, torch.cuda.device_count() = {have}: not starting "
              "(nothing was launched; run with --gpus <= that, or on a node that has them)", file=sys.stderr, flush=True)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("DLDKD_COMM_DEADLINE_S", "300")
    cmd = launch_command(a.gpus, argv, free_port())
    print("bench.py: starting " + " ".join(cmd), file=sys.stderr, flush=True)
    return relay_child(cmd, timeout_s=a.launch_timeout, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--launch-timeout", type=float, default=3600.0,
                    help="N > 1 self-launch only: seconds before the launched ranks are killed (a hang must not outlive the lease)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # nobody started our ranks: do it ourselves, as a child process, BEFORE anything in this process touches a GPU
        sys.exit(self_launch(a, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # (child process, before this one touches the GPU; DLDKD_BENCH_NO_MFMA_PROBE=1: skipped - the PMC passes of
    #  tools/collect_profiles_r06.sh profile THIS script's loop and want no second program under the profiler)
    sustained = mfma_sustained() if (world == 1 and rank == 0 and os.environ.get("DLDKD_BENCH_NO_MFMA_PROBE") != "1") else None
    if world != a.gpus and not (world == 1 and os.environ.get("DLDKD_BENCH_FORCE_DIST") == "1"):
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: start one rank per GPU (or leave WORLD_SIZE unset and "
                         "let bench.py start them)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} has LOCAL_RANK {local_rank} but this node has {torch.cuda.device_count()} GPUs")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    comm = None
    force_dist = os.environ.get("DLDKD_BENCH_FORCE_DIST") == "1"     # test hook: distributed code path with one rank
    if world > 1 or force_dist:
        # one RCCL communicator over the ranks, driven through the C ABI (dldkd_amd.comm.RcclComm): every collective is one enqueue
        # on the caller's stream - no torch.distributed process group, no watchdog thread polling events beside the graph captures
        # of the training step.  The rendezvous id travels over the env:// TCP store torch.distributed.run set up.
        from dldkd_amd import comm as dcomm
        comm = dcomm.init_rccl_from_env(dev)

    from dldkd_amd import native, scoring
    native.lib()   # fail loudly before anything else if the HIP library is missing
    metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    out = None

    if comm is not None:
        # ---- N ranks (or the one-rank test hook): the sharded step, self-verified, then the C4 workload the same way
        r2 = run_sharded(C2, dev, rank, world, comm, a.steps, a.warmup)
        r4 = run_sharded(C4, dev, rank, world, comm, max(min(a.steps, 20), 1), max(min(a.warmup, 3), 1)) if not a.no_extras else None
        r5 = run_c5_ddp(dev, rank, world, comm, max(min(a.steps, 20), 2), max(min(a.warmup, 5), 3)) if not a.no_extras else None
        if rank == 0:
            achieved = r2["flops_per_step_all_ranks"] / (r2["ms_per_step"] * 1e-3) / 1e12
            out = {
                "metric": metric, "value": r2["value"], "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": r2["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "bf16", "data": "synthetic",
                "config": {"workload": C2["workload"], "n_queries": NQ, "n_videos": NV, "max_clips": L,
                           "clip_len": f"U{{{LEN_LO}..{L}}}", "hidden": D, "branches": NB, "fusion": list(W_FUSE),
                           "parallelism": f"gallery sharded x{world} by video ({r2['shard_videos']} per rank): one scorer launch per step, "
                                          f"{r2['n_ranges']} query ranges completing in order, the all_gather of each range overlapped "
                                          "with the scoring of the next",
                           "step": "pack queries + simpool (sim + key-clip max-pool) + 0.7/0.3 fusion + all_gather",
                           "gallery_pack_ms_untimed": r2["gallery_pack_ms_untimed"]},
                "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS * world, "unit": "TFLOP/s",
                             "frac": achieved / (PEAK_BF16_TFLOPS * world), "traffic": None,
                             "kernel": "simpool_eval16p_kernel on every rank (one launch, query ranges in order) + per-range finish and "
                                       "all_gather on a side stream; `achieved` = all ranks' algorithmic flops / the step's wall time "
                                       "(max over ranks), `peak` = world x 2.5 PF",
                             "kernel_ms": r2["step_stream_ms"], "algorithmic_flops_per_launch": r2["flops_per_step_all_ranks"] / world},
                "cpu_baseline": None,
                # self-verification of the distributed path (the same synthetic gallery and queries for every N):
                "recall_hip": r2["recall_hip"], "assembled_max_abs_diff": r2["assembled_max_abs_diff"],
                "assembled_check": r2["assembled_check"],
                "recall_expected_n1": RECALL_N1, "recall_matches_n1": r2["recall_hip"] == RECALL_N1,
            }
            if r4 is not None:
                out["extras"] = {"c4_sharded": {
                    "workload": C4["workload"], "n_queries": C4["nq"], "n_videos": C4["nv"], "clips": 128, "n_gpus": world,
                    "ms_per_step": r4["ms_per_step"], "pairs_per_s": r4["value"], "n_ranges": r4["n_ranges"],
                    "shard_videos": r4["shard_videos"],
                    "algorithmic_TFLOPs_all_ranks": r4["flops_per_step_all_ranks"] / (r4["ms_per_step"] * 1e-3) / 1e12,
                    "recall_hip": r4["recall_hip"], "assembled_max_abs_diff": r4["assembled_max_abs_diff"],
                    "assembled_check": r4["assembled_check"], "recall_expected_n1": RECALL_C4_N1,
                    "recall_matches_n1": r4["recall_hip"] == RECALL_C4_N1},
                    "c5_ddp": r5}
        comm.barrier()
    else:
        # ---- one GPU: the headline line with the kernel's roofline, the CPU baseline and the extras
        gs, mask, lens, qs, gt = synth_shard(dev, C2, 0, NV)
        t0 = time.perf_counter()
        pg = scoring.pack_gallery(gs, mask)            # resident bf16 gallery (outside the timed region)
        torch.cuda.synchronize()
        pack_gallery_ms = (time.perf_counter() - t0) * 1e3
        keep_fp32 = not a.no_cpu_baseline
        if not keep_fp32:
            gs = None
        ws = torch.empty(native.lib().dldkd_simpool_eval_workspace_bytes(NQ, NV, NB), dtype=torch.uint8, device=dev)
        flops_launch = 2.0 * D * NB * NQ * float(lens.sum().item())   # algorithmic: valid clips only
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]

        def step(i=None):
            pq = scoring.pack_queries(qs)                                  # F.normalize + bf16 (model.py:318)
            if i is not None:
                ev[i][0].record()
            scoring.simpool_partials(pq, pg, ws)                           # the dominant kernel
            if i is not None:
                ev[i][1].record()
            fused, _, _ = scoring.simpool_finish(ws, pq, pg, W_FUSE)       # (NQ, NV) fp32
            return fused

        for _ in range(a.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            fused = step(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kern_ms = sum(s_.elapsed_time(e) for s_, e in ev) / max(a.steps, 1)
        achieved = flops_launch / (kern_ms * 1e-3) / 1e12
        # HBM bytes per launch of the dominant kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes,
        # KiB units; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md section HBM) from the committed
        # profile of this same workload.  Not re-measured live (PMC needs the profiler).
        traffic, traffic_src = None, None
        for pmc in [os.path.join(ROOT, "profiles", r_, "pmc_simpool.json") for r_ in ("r06", "r05")] + [os.path.join(ROOT, "profiles", r_, "pmc_simpool", "summary.json")
                                                                                  for r_ in ("r04", "r03", "r02")]:
            rnd = os.path.relpath(pmc, os.path.join(ROOT, "profiles")).split(os.sep)[0]
            if os.path.exists(pmc):
                d = json.load(open(pmc))
                traffic = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
                traffic_src = f"{os.path.relpath(pmc, ROOT)} (rocprofv3 --pmc on `bench.py --no-extras --no-cpu-baseline` itself since round 6, separate passes; 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE, KiB units)"
                break
        out = {
            "metric": metric, "value": NQ * NV * a.steps / dt, "unit": "pairs/s", "n_gpus": 1,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": C2["workload"],
                       "n_queries": NQ, "n_videos": NV, "max_clips": L, "clip_len": f"U{{{LEN_LO}..{L}}}",
                       "hidden": D, "branches": NB, "fusion": list(W_FUSE), "parallelism": "1 GPU",
                       "step": "pack queries + simpool (sim + key-clip max-pool) + 0.7/0.3 fusion",
                       "gallery_pack_ms_untimed": round(pack_gallery_ms, 2)},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src, "kernel_ms": kern_ms,
                         "kernel": "simpool_eval16p_kernel" if pg.scorer_waves() != pg.nv else "simpool_eval16_kernel",
                         "scorer_waves": pg.scorer_waves(),
                         "algorithmic_flops_per_launch": flops_launch},
        }
        if sustained and sustained.get("random_16x16x32"):
            out["roofline"]["sustained_register_operand_mfma_TFLOPs"] = sustained
            out["roofline"]["frac_of_sustained_random_data"] = achieved / sustained["random_16x16x32"]
        out["recall_hip"] = dict(zip(("R@1", "R@5", "R@10", "R@100"), recalls(fused, gt)))
        out["recall_expected_n1"] = RECALL_N1          # what every N must reproduce (same gallery, same queries)
        if keep_fp32:
            cb, err = cpu_baseline(gs, mask, qs, fused)
            out["cpu_baseline"] = cb
            out["parity_max_abs_err_vs_oracle_sample"] = err
            out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
        else:
            out["cpu_baseline"] = None
        if not a.no_extras:
            del gs, fused
            torch.cuda.empty_cache()
            out["extras"] = extras(dev)
    if rank == 0:
        # the JSON line must be the LAST line of stdout: RCCL / the HIP runtime write banners through C stdio, which is
        # flushed at exit - after Python's own prints - unless it is drained first
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:   # noqa: BLE001
            pass
        print(json.dumps(out), flush=True)
    if comm is not None:
        # every rank has passed the barrier above and drained its streams: free the communicator like any other resource
        comm.destroy()


if __name__ == "__main__":
    main()
