"""Time the C3 training step (TVR: 128 videos / 640 queries, soft labels, hard negatives) on the GPU box."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("dl-dkd_amd", "tests/golden"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, synth
from dldkd_amd.model import DLDKD
from dldkd_amd.optimization import BertAdam
DEV = "cuda:0"
drop = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
from dldkd_amd import ops
ops.set_gemm_precision(prec)
cfg = types.SimpleNamespace(visual_input_size=3072, query_input_size=768, inheritance_hidden=384, exploration_hidden=384,
                            max_ctx_l=128, max_desc_l=30, input_drop=drop, drop=drop, n_heads=4, initializer_range=0.02,
                            margin=0.1, use_hard_negative=True, hard_pool_size=20, label_style="soft")
opt_ = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                             collection="tvr", alpha=0.8, belta=0.8)
torch.manual_seed(0)
m = DLDKD(cfg, opt_).to(DEV).train()
opt = BertAdam([{"params": list(m.parameters()), "weight_decay": 0.01}], lr=3e-4, warmup=0.01, t_total=1000)
batch = synth.make_train_batch(3, nv=128, caps=5, L=128, len_lo=24, dv=3072, dq=768)
batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
def step():
    opt.zero_grad(); loss, _ = m(batch); loss.backward(); opt.step(); return loss
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"C3 train step ({prec} GEMMs, dropout {drop}): {dt*1e3:.2f} ms/step  loss {float(l):.4f}")
