"""Flakiness hunt: one-rank RCCL group + train() with / without the captured step, destroy_process_group at the end."""
import os, sys, types, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("dl-dkd_amd", "tests/golden", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import faulthandler; faulthandler.enable()
import torch
import torch.distributed as dist
from test_train_loop_gpu import TinySet, L
from dldkd_amd.model import DLDKD
from dldkd_amd import train as T
DEV = "cuda:0"
graph = sys.argv[1] == "1"
ds = TinySet()
cfg = types.SimpleNamespace(visual_input_size=256, query_input_size=128, inheritance_hidden=384, exploration_hidden=384,
                            max_ctx_l=128, max_desc_l=30, input_drop=0.1, drop=0.1, n_heads=4, initializer_range=0.02,
                            margin=0.1, use_hard_negative=False, hard_pool_size=5, label_style="soft")
opt = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                            collection="tiny", alpha=0.8, belta=0.8, device=torch.device(DEV), bsz=8, pin_memory=False,
                            num_workers=0, lr=1e-3, wd=0.01, lr_warmup_proportion=0.05, n_epoch=5, max_es_cnt=10,
                            hard_negative_start_epoch=0, hard_pool_size=5, distill_loss_decay="exp", exponential_k=0.95,
                            selfDistil_sigmoid_k=800, alpha_decay="sigmoid", belta_decay="sigmoid", grad_clip=-1,
                            eval_context_bsz=16, eval_query_bsz=50, eval_untrained=True, graph_step=graph,
                            ckpt_filepath=os.path.join(tempfile.mkdtemp(), "m.ckpt"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29591", RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", device_id=torch.device(DEV))
T.DDP_MIN_WORLD = 1
torch.manual_seed(0)
m = DLDKD(cfg, opt)
h = T.train(m, ds, L(ds.videos()), L(ds.texts()), opt)
torch.cuda.synchronize()
print("trained", h[-1][2], flush=True)
dist.destroy_process_group()
print("destroyed ok", flush=True)
