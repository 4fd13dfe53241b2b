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
for mode in ("fp32", "bf16"):
    ops.set_gemm_precision(mode)
    m.fast_input_proj = mode == "bf16"
    with torch.no_grad():
        ev.eval_epoch(m, synth.ListDataset(list(vids)), synth.ListDataset(list(txts)), opt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s = ev.eval_epoch(m, synth.ListDataset(list(vids)), synth.ListDataset(list(txts)), opt)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"eval_epoch {mode}: {dt:.3f} s wall for 2000 videos / 6000 queries (SumR {s:.1f})")

if os.environ.get("PROFILE"):
    import cProfile, pstats
    ops.set_gemm_precision("bf16"); m.fast_input_proj = True
    pr = cProfile.Profile()
    with torch.no_grad():
        pr.enable()
        ev.eval_epoch(m, synth.ListDataset(list(vids)), synth.ListDataset(list(txts)), opt)
        pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
