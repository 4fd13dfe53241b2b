"""cProfile of the host side of one C5-shaped training step (small shapes: launch-bound)."""
import os, sys, types, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("dl-dkd_amd", "tests/golden"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, synth
from dldkd_amd.model import DLDKD
from dldkd_amd.optimization import BertAdam
from dldkd_amd import ops
DEV = "cuda:0"
cfg = types.SimpleNamespace(visual_input_size=1024, query_input_size=1024, inheritance_hidden=384, exploration_hidden=384,
                            max_ctx_l=128, max_desc_l=30, input_drop=0.15, drop=0.15, n_heads=4, initializer_range=0.02,
                            margin=0.1, use_hard_negative=True, hard_pool_size=20, label_style="soft")
opt_ = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                             collection="charades", alpha=0.8, belta=0.8)
torch.manual_seed(0)
m = DLDKD(cfg, opt_).to(DEV).train()
opt = BertAdam([{"params": list(m.parameters()), "weight_decay": 0.01}], lr=2.4e-4, warmup=0.01, t_total=1000)
caps5 = sorted([3] + [2] * 127, reverse=True)
batch = synth.make_train_batch(5, nv=128, caps=caps5, L=64, len_lo=8, dv=1024, dq=1024)
batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
ops.set_gemm_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
def step():
    opt.zero_grad(); loss, _ = m(batch); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
