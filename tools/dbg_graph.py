import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("dl-dkd_amd", "tests/golden"):
    sys.path.insert(0, os.path.join(ROOT, p))
import faulthandler; faulthandler.enable()
import torch, synth
from dldkd_amd import ops, train as T
from dldkd_amd.model import DLDKD
from dldkd_amd.optimization import BertAdam
DEV = "cuda:0"
hard = sys.argv[1] == "1"; drop = float(sys.argv[2]); dv = int(sys.argv[3]); L = int(sys.argv[4]); nv = int(sys.argv[5])
cfg = types.SimpleNamespace(visual_input_size=dv, query_input_size=128, inheritance_hidden=384, exploration_hidden=384,
                            max_ctx_l=128, max_desc_l=30, input_drop=drop, drop=drop, n_heads=4, initializer_range=0.02,
                            margin=0.1, use_hard_negative=hard, hard_pool_size=20, label_style="soft")
mopt = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                             collection="tvr", alpha=0.8, belta=0.8)
topt = types.SimpleNamespace(grad_clip=-1)
b = synth.make_train_batch(70, nv=nv, caps=2, L=L, len_lo=3, dv=dv, dq=128)
b = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items()}
torch.manual_seed(11)
m = DLDKD(cfg, mopt).to(DEV).train()
opt = BertAdam([{"params": list(m.parameters()), "weight_decay": 0.01}], lr=2e-3, warmup=0.1, t_total=40)
st = T.GraphedTrainStep(m, opt, topt)
for it in range(4):
    _, d = st(b)
    print(it, d["loss_overall"], flush=True)
print("ok", st.replays, st.captures)
