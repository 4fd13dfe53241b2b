"""Stage timings of the eval path at TVR scale (GPU box): gallery encode from raw i3d-dim features -> towers ->
resident bf16 gallery (streamed, as eval_epoch does), query encode (super-batches), scoring, ranking.
Synthetic features are generated on the device (one 200-video batch re-used: what is timed is the GPU work of
eval_epoch; the DataLoader / BigFile side is row f3 and is not part of this number)."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dl-dkd_amd"))
import torch


def stage_times(nv=21793, nq=10895, mode="fp32", dev="cuda:0", B=None):
    from dldkd_amd.model import DLDKD
    from dldkd_amd import scoring, ops, eval as ev
    cfg = types.SimpleNamespace(visual_input_size=3072, query_input_size=768, inheritance_hidden=384, exploration_hidden=384,
                                max_ctx_l=128, max_desc_l=30, input_drop=0.2, drop=0.2, n_heads=4, initializer_range=0.02,
                                margin=0.1, use_hard_negative=True, hard_pool_size=20, label_style="soft")
    opt_ = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                 collection="tvr", alpha=0.8, belta=0.8)
    torch.manual_seed(0)
    m = DLDKD(cfg, opt_).to(dev).eval()
    if mode in ("fast", "resident"):         # K4 input projection + every tower GEMM on bf16 MFMA
        m.fast_input_proj = True
        ops.set_gemm_precision("bf16")
    gen = torch.Generator(device=dev).manual_seed(1)

    def sync():
        torch.cuda.synchronize()
        return time.perf_counter()
    out = {"mode": mode, "n_videos": nv, "n_queries": nq}
    from dldkd_amd import eval as ev
    try:
        with torch.no_grad():
            B = B or ev.CONTEXT_SUPER_BATCH          # videos per encode call (eval.py groups loader batches up to this)
            L = 128
            feats = torch.nn.functional.normalize(torch.randn(B, L, 3072, generator=gen, device=dev), dim=-1)
            lens = torch.randint(24, L + 1, (B,), generator=gen, device=dev)
            mask = (torch.arange(L, device=dev).unsqueeze(0) < lens.unsqueeze(1)).float()
            feats = feats * mask.unsqueeze(-1)
            lens_host = lens.cpu().numpy()                       # eval.py has them from the loader's CPU mask
            if mode == "resident":
                # what eval_epoch runs in throughput mode since round 3b: the gallery's raw features live on the device as a
                # ragged bf16 table + per-row LayerNorm statistics (filled once, from the first pass's loader batches); an epoch's
                # gallery encode is K4b over the whole table + the fused tower kernel over all videos
                res = ev.ResidentGallery(3072, torch.device(dev))
                t0 = sync()
                done = 0
                while done < nv:
                    n = min(B, nv - done)
                    res.table.append(feats[:n], lens_host[:n])
                    done += n
                t1 = sync()
                out["resident_fill_once"] = {"s": t1 - t0, "table_GB": res.table.nbytes() / 1e9, "clips": res.table.rows,
                                             "padded_fp32_batches_GB": nv * L * 3072 * 4 / 1e9}
                res.complete = True
                res.plan(torch.device(dev))
                out["chunks"] = len(res.chunks)
                for _ in range(2):
                    wpk = scoring.GalleryPacker(nv, L, 2, torch.device(dev))
                    m.encode_resident_into(wpk, res)
                    wpk.finish()
                del wpk
                pk = scoring.GalleryPacker(nv, L, 2, torch.device(dev))
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                t0 = sync()
                evs[0].record()
                m.encode_resident_into(pk, res)
                evs[1].record()
                th = time.perf_counter()
                pg = pk.finish()
                evs[2].record()
                t1 = sync()
                out["gallery_encode_and_pack_s"] = t1 - t0
                out["gallery_detail"] = {"host_enqueue_ms": (th - t0) * 1e3, "gpu_encode_ms": evs[0].elapsed_time(evs[1]),
                                         "gpu_finish_order_ms": evs[1].elapsed_time(evs[2])}
                out["gallery_videos_per_sec"] = nv / (t1 - t0)
                del res
            elif mode == "fast":                                 # eval.py cuts super-batches where the tower kernel's workgroups fill whole rounds
                B = ev._take_for_budget(lens_host, ev.TOWER_ITEM_BUDGET)
                out["videos_per_super_batch"] = int(B)

            if mode != "resident":
                def encode(pk_, n):                                  # what compute_context_info's flush() does per super-batch
                    if not (mode == "fast" and m.encode_context_into(pk_, feats[:n], mask[:n], lens_host=lens_host[:n])):
                        gi, ge = m.encode_context(feats[:n], mask[:n])
                        pk_.add([gi, ge], mask[:n])
                for _ in range(2):                                   # warm-up: kernel modules, allocator pools, torch's lazy sort
                    wpk = scoring.GalleryPacker(B, L, 2, torch.device(dev))
                    encode(wpk, B)
                    wpk.finish()
                del wpk
                pk = scoring.GalleryPacker(nv, L, 2, torch.device(dev))   # 2 x 2.1 GB: first-touch hipMalloc is not GPU work
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                t0 = sync()
                evs[0].record()
                done = 0
                prof = None
                if os.environ.get("E2E_HOSTPROF"):                   # where does the host spend the enqueue loop?  (stderr)
                    import cProfile
                    prof = cProfile.Profile()
                    prof.enable()
                while done < nv:
                    n = min(B, nv - done)
                    encode(pk, n)
                    done += n
                if prof is not None:
                    prof.disable()
                    import pstats
                    pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(14)
                evs[1].record()
                th = time.perf_counter()
                pg = pk.finish()
                evs[2].record()
                t1 = sync()
                out["gallery_encode_and_pack_s"] = t1 - t0
                out["gallery_detail"] = {"host_enqueue_loop_ms": (th - t0) * 1e3, "gpu_loop_ms": evs[0].elapsed_time(evs[1]),
                                         "gpu_finish_order_ms": evs[1].elapsed_time(evs[2])}
                out["gallery_videos_per_sec"] = nv / (t1 - t0)
            SB = ev.QUERY_SUPER_BATCH
            words = torch.nn.functional.normalize(torch.randn(SB, 30, 768, generator=gen, device=dev), dim=-1)
            wl = torch.randint(5, 31, (SB,), generator=gen, device=dev)
            wmask = (torch.arange(30, device=dev).unsqueeze(0) < wl.unsqueeze(1)).float()
            m.encode_query(words, wmask)
            t0 = sync()
            qi, qe = [], []
            for lo in range(0, nq, SB):
                n = min(SB, nq - lo)
                a, b = m.encode_query(words[:n], wmask[:n])
                qi.append(a)
                qe.append(b)
            t1 = sync()
            out["query_encode_s"] = t1 - t0
            qs = [torch.cat(qi), torch.cat(qe)]
            m.pooled_scores(qs, pg)
            t0 = sync()
            fused, s0, s1 = m.pooled_scores(qs, pg, want_branches=True)
            t1 = sync()
            out["scoring_3_matrices_s"] = t1 - t0
            m3 = t1 - t0
            gt = {q: [q % nv] for q in range(nq)}
            ev.gt_ranks_gpu(fused, gt)
            t0 = sync()
            csr = ev.gt_csr(gt, nq, fused.device)          # once per epoch, as eval_epoch does
            for sc in (s0, s1, fused):                     # eval_epoch ranks all three (eval.py:246-254)
                rb, rf = ev.gt_ranks_gpu(sc, gt, csr)
                rb.cpu(), rf.cpu()
            t1 = sync()
            out["ranking_3_matrices_s"] = t1 - t0
            r3 = t1 - t0
            # what eval_epoch runs since round 2: scorer partial planes -> ranks of all three score kinds in one pass over
            # the planes (no (Nq, Nv) matrix written or read), ranks to the host
            pq = scoring.pack_queries(qs)
            ws = scoring.simpool_partials(pq, pg)
            scoring.rank_partials(ws, pq, pg, csr[0], csr[1])
            t0 = sync()
            pq = scoring.pack_queries(qs)
            ws = scoring.simpool_partials(pq, pg, ws)
            t1 = sync()
            ranks = scoring.rank_partials(ws, pq, pg, csr[0], csr[1]).cpu()
            t2 = sync()
            out["scoring_partials_only_s"] = t1 - t0
            out["ranking_from_partials_s"] = t2 - t1
            del out["scoring_3_matrices_s"], out["ranking_3_matrices_s"]
            out["total_s"] = sum(v for k, v in out.items() if k.endswith("_s"))
            out["matrix_path"] = {"scoring_3_matrices": m3, "ranking_3_matrices": r3}       # round-1 path, kept for comparison
    finally:
        ops.set_gemm_precision("fp32")
    return out


if __name__ == "__main__":
    NV = int(sys.argv[1]) if len(sys.argv) > 1 else 21793
    NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 10895
    BB = int(os.environ.get("E2E_BATCH", "0")) or None
    for mode in (sys.argv[3:] or ["fp32", "fast"]):
        print(stage_times(NV, NQ, mode, B=BB))
