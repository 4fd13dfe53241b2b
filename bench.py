#!/usr/bin/env python3
"""bench.py - headline benchmark: query x video pairs scored / s on the TVR full gallery (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

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


def synth_shard(dev, v_lo, v_hi, seed=2):
    """Encoded-gallery shard [v_lo, v_hi) for both branches + planted queries (same on every rank)."""
    lens_all = torch.randint(LEN_LO, L + 1, (NV,), generator=torch.Generator().manual_seed(seed)).to(dev)
    n_loc = v_hi - v_lo
    gs = []
    for b in range(NB):
        gen = torch.Generator(device=dev).manual_seed(1000 * seed + 10 * v_lo + b)
        gs.append(torch.randn(n_loc, L, D, generator=gen, device=dev))
    lens = lens_all[v_lo:v_hi]
    mask = (torch.arange(L, device=dev).unsqueeze(0) < lens.unsqueeze(1)).float()
    # queries: planted on a valid clip of video (q mod NV) when that video is local, plain noise otherwise
    qgen = torch.Generator(device=dev).manual_seed(seed + 7)
    gt = torch.arange(NQ, device=dev) % NV
    lstar = (torch.rand(NQ, generator=qgen, device=dev) * lens_all[gt].float()).long().clamp(max=L - 1)
    qs = []
    for b in range(NB):
        noise = torch.randn(NQ, D, generator=qgen, device=dev)
        local = (gt >= v_lo) & (gt < v_hi)
        base = torch.zeros(NQ, D, device=dev)
        base[local] = gs[b][(gt[local] - v_lo), lstar[local]]
        qs.append(base + SIGMA[b] * noise)
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
    torch.set_num_threads(all_cores)
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
            for prec in ("fp32", "bf16"):
                r = bench_train.run(cfgname, prec, 0.2 if cfgname == "c3" else 0.15, steps=30, warmup=10, dev=str(dev))
                out[f"{key}_train_step_{prec}"] = r
                out[f"{key}_train_step_ms" + ("_bf16" if prec == "bf16" else "") if key == "c3" else f"c5_train_step_ms_{prec}"] = \
                    r["graph"]["stream_ms_median"]
        out["c3_train_step_config"] = "TVR: 128 videos / 640 queries, L<=128, label_style=soft, hard negatives, dropout 0.2, " \
                                      "zero_grad + forward + backward + fused BertAdam; fp32 = parity mode (fp32-grade GEMMs: three bf16 " \
                                      "planes per operand, losses within 1e-4 of the reference), bf16 = every GEMM on bf16 MFMA with fp32 " \
                                      "accumulation (fp32 master weights / activations, losses 2e-2); *_ms = hipGraph-replayed step, median"
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
                ops.in_proj_bf16(xk, fold)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
            evs[0].record()
            for i in range(10):
                ops.in_proj_bf16(xk, fold)
                evs[i + 1].record()
            torch.cuda.synchronize()
            return sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(10))[5]
        ms = k4_ms()
        byts = 400000 * 3072 * 4 + 400000 * 768 * 4 + 768 * 3072 * 2
        out["k4_in_proj_roofline"] = {"bound": "hbm", "achieved": byts / ms / 1e6, "peak": 8000.0, "unit": "GB/s",
                                      "frac": byts / ms / 1e6 / 8000.0, "kernel": "in_proj_rows128_kernel (dldkd_in_proj_bf16_rows128)", "kernel_ms": ms,
                                      "shape": "400000 rows x 3072 fp32 -> 2 x 384 fp32", "timing": "median of 10 launches after 3 warm-ups"}
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    sustained = mfma_sustained() if (world == 1 and rank == 0) else None    # child process, before this one touches the GPU
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run --nproc-per-node N (one rank per GPU)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist = None
    force_dist = os.environ.get("DLDKD_BENCH_FORCE_DIST") == "1"     # test hook: distributed code path with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL kernels need CUs too, and the scorer parks a 512-register wave on every SIMD: give the collective's
        # stream high priority so its workgroups are dispatched first whenever a scorer workgroup retires
        try:
            pg_opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            dist.init_process_group("nccl", device_id=dev, pg_options=pg_opts)
        except Exception:   # noqa: BLE001  (older/newer torch without this option)
            dist.init_process_group("nccl", device_id=dev)

    from dldkd_amd import native, scoring
    native.lib()   # fail loudly before anything else if the HIP library is missing

    shard = (NV + world - 1) // world
    v_lo, v_hi = min(rank * shard, NV), min((rank + 1) * shard, NV)
    gs, mask, lens, qs, gt = synth_shard(dev, v_lo, v_hi)
    n_loc = v_hi - v_lo
    if n_loc < shard:   # pad the last shard with 1-clip zero videos so all_gather blocks are equal
        pad = shard - n_loc
        gs = [torch.cat([g, torch.zeros(pad, L, D, device=dev)]) for g in gs]
        mask = torch.cat([mask, torch.zeros(pad, L, device=dev)])
        mask[n_loc:, 0] = 1.0
    t0 = time.perf_counter()
    pg = scoring.pack_gallery(gs, mask)            # resident bf16 gallery (outside the timed region)
    torch.cuda.synchronize()
    pack_gallery_ms = (time.perf_counter() - t0) * 1e3
    keep_fp32 = (world == 1 and not a.no_cpu_baseline)
    if not keep_fp32:
        gs = None
    ws = torch.empty(native.lib().dldkd_simpool_eval_workspace_bytes(NQ, shard, NB), dtype=torch.uint8, device=dev)
    overlap = None
    if world > 1 or force_dist:
        from dldkd_amd import dist as ddist
        # ONE scorer launch over all queries, grid [query range][branch][4 videos]; >= 4 ranges so that the all-gather of
        # range r (RCCL stream, parked on the range's arrival counter) runs under the scoring of ranges r+1..
        n_ranges, per_range = scoring.plan_query_split(NQ, shard, NB, min_split=4)
        backend = ddist.HipShardBackend(qs, pg, n_ranges, W_FUSE)
        overlap = ddist.OverlappedShardScorer(backend, ddist.query_ranges(NQ, n_ranges, per_range), shard, dev)
    flops_launch = 2.0 * D * NB * NQ * float(lens.sum().item())   # algorithmic: valid clips only

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]

    def step(i=None):
        if overlap is not None:
            # one launch; the RCCL all-gather of query range r overlaps the scoring of ranges r+1..
            if i is not None:
                ev[i][0].record()
            overlap.step()
            if i is not None:
                ev[i][1].record()
            return overlap.local[-1]
        pq = scoring.pack_queries(qs)                                  # F.normalize + bf16 (model.py:318)
        if i is not None:
            ev[i][0].record()
        scoring.simpool_partials(pq, pg, ws)                           # the dominant kernel
        if i is not None:
            ev[i][1].record()
        fused, _, _ = scoring.simpool_finish(ws, pq, pg, W_FUSE)       # (NQ, shard) fp32
        return fused

    for _ in range(a.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        fused = step(i)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    kern_ms = sum(s.elapsed_time(e) for s, e in ev) / max(a.steps, 1)

    out = None
    if rank == 0:
        metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        achieved = flops_launch / (kern_ms * 1e-3) / 1e12
        # HBM bytes per launch of the dominant kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes,
        # KiB units; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md section HBM) from the committed
        # profile of this same workload.  Not re-measured live (PMC needs the profiler), so N>1 reports null.
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r02", "pmc_simpool", "summary.json")
        if world == 1 and os.path.exists(pmc):
            d = json.load(open(pmc))
            traffic = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
            traffic_src = "profiles/r02/pmc_simpool/summary.json (rocprofv3 --pmc, separate passes; 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE, KiB units)"
        out = {
            "metric": metric, "value": NQ * NV * a.steps / dt, "unit": "pairs/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "C2: TVR full eval gallery text->video scoring (configs[1])",
                       "n_queries": NQ, "n_videos": NV, "max_clips": L, "clip_len": f"U{{{LEN_LO}..{L}}}",
                       "hidden": D, "branches": NB, "fusion": list(W_FUSE),
                       "parallelism": "1 GPU" if world == 1 else f"gallery sharded x{world}: one scorer launch, {len(overlap.bounds)} query ranges completing in order, all_gather of each range overlapped with the scoring of the next",
                       "step": "pack queries + simpool (sim + key-clip max-pool) + 0.7/0.3 fusion"
                               + (" + all_gather" if world > 1 else ""),
                       "gallery_pack_ms_untimed": round(pack_gallery_ms, 2)},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src,
                         "kernel": "simpool_eval16_kernel" if overlap is None else
                                   "simpool_eval16_kernel (one launch, query ranges in order) + per-range finish and all_gather on a side stream; kernel_ms spans the whole step",
                         "kernel_ms": kern_ms,
                         "algorithmic_flops_per_launch": flops_launch},
        }
        if sustained and sustained.get("random_16x16x32"):
            out["roofline"]["sustained_register_operand_mfma_TFLOPs"] = sustained
            out["roofline"]["frac_of_sustained_random_data"] = achieved / sustained["random_16x16x32"]
        if world == 1 and overlap is None:
            out["recall_hip"] = dict(zip(("R@1", "R@5", "R@10", "R@100"), recalls(fused, gt)))
        if overlap is not None and world == 1:     # test hook: the chunked path must reproduce the one-launch matrix
            pq = scoring.pack_queries(qs)
            scoring.simpool_partials(pq, pg, ws)
            ref, _, _ = scoring.simpool_finish(ws, pq, pg, W_FUSE)
            out["force_dist_max_abs_diff"] = (overlap.assemble(NV) - ref[:, :NV]).abs().max().item()
        if keep_fp32 and overlap is None:
            cb, err = cpu_baseline(gs, mask, qs, fused)
            out["cpu_baseline"] = cb
            out["parity_max_abs_err_vs_oracle_sample"] = err
            out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
        else:
            out["cpu_baseline"] = None
        if world == 1 and not a.no_extras and overlap is None:
            del gs, fused
            torch.cuda.empty_cache()
            out["extras"] = extras(dev)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
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


if __name__ == "__main__":
    main()
