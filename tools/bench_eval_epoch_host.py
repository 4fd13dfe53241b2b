"""Wall time of eval_epoch INCLUDING the host side (DataLoader, collate, H2D) on an in-memory synthetic dataset:
2000 videos x U{8..64} clips x 1024-d, 3 captions each (ActivityNet-like dims).  The GPU work is a small part of it."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("dl-dkd_amd", "tests/golden"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, synth
from dldkd_amd.model import DLDKD
from dldkd_amd import eval as ev, ops
cfg = types.SimpleNamespace(visual_input_size=1024, query_input_size=1024, inheritance_hidden=384, exploration_hidden=384,
                            max_ctx_l=128, max_desc_l=30, input_drop=0.2, drop=0.2, n_heads=4, initializer_range=0.02,
                            margin=0.1, use_hard_negative=True, hard_pool_size=20, label_style="soft")
opt_ = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                             collection="activitynet", alpha=0.8, belta=0.8)
torch.manual_seed(0)
m = DLDKD(cfg, opt_).to("cuda:0").eval()
vids, txts = synth.make_eval_sets(3, nv=2000, caps=3, len_lo=8, len_hi=64, dv=1024, dq=1024)
opt = types.SimpleNamespace(eval_context_bsz=200, eval_query_bsz=50, num_workers=0, pin_memory=False, device=torch.device("cuda:0"),
                            double_branch=True)
dsv, dst = synth.ListDataset(list(vids)), synth.ListDataset(list(txts))
for mode in ("parity", "throughput"):
    opt.eval_precision = mode
    with torch.no_grad():
        opt.eval_feature_cache = False
        ev.eval_epoch(m, dsv, dst, opt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s = ev.eval_epoch(m, dsv, dst, opt)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        opt.eval_feature_cache = True                      # the default: raw features of the first pass stay on the device
        ev.clear_feature_cache()
        ev.eval_epoch(m, dsv, dst, opt)                     # fills the cache
        ev.eval_epoch(m, dsv, dst, opt)                     # first replay: the allocator still grows its pools
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s2 = ev.eval_epoch(m, dsv, dst, opt)                # steady state
        torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
    print(f"eval_epoch {mode}: {dt:.3f} s wall from host features, {dt2:.3f} s with the device feature cache, 2000 videos / 6000 queries "
          f"(SumR {s:.1f} / {s2:.1f})")

if os.environ.get("PROFILE"):
    import cProfile, pstats
    opt.eval_precision = "throughput"; opt.eval_feature_cache = os.environ.get("PROFILE") == "cache"
    pr = cProfile.Profile()
    with torch.no_grad():
        ev.eval_epoch(m, dsv, dst, opt)
        pr.enable()
        ev.eval_epoch(m, dsv, dst, opt)
        torch.cuda.synchronize()
        pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
