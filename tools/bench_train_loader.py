"""End-to-end training throughput INCLUDING the data path, on an in-memory synthetic training set of TVR shapes (N videos x
U{24..128} clips x 3072-d student / 512-d teacher features, 5 captions of 5..30 words x 768-d each): train.train_epoch with
  host:    DataLoader + collate_train (pad on the host) + .to(device) every step, as the reference does,
  device:  data.DeviceTrainSet / DeviceTrainLoader (items read once into ragged device tables, batches gathered by a kernel).
Prints steps/s (wall) of the second epoch of each and the one-time cost of building the device tables."""
import json, os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("dl-dkd_amd", "tests/golden"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import torch


class SynthTrainSet(torch.utils.data.Dataset):
    """config "c3": TVR shapes (24..128 clips x 3072, 5 captions of 5..30 words x 768 per video); "c5": Charades shapes (8..64 clips x
    1024, 1..6 captions per video - mean 2.3, as Charades-STA's 12,408 / 5,338 - of 4..12 words x 1024): every batch of 128 videos
    then has its own number of queries; "anet": ActivityNet shapes (16..128 clips x 1024, 1..10 captions per video, mean 3.7)."""

    def __init__(self, n, seed=0, config="c3", pool=None):
        """pool: generate that many distinct items and let item i be item i % pool with its own id (a 17,435-video TVR-sized set
        costs 21 GB and a minute of random numbers otherwise; the device tables still hold every item's rows)."""
        rs = np.random.RandomState(seed)
        self.items = []
        self.n, n = n, (min(n, pool) if pool else n)
        dv, dq, l_lo, l_hi, w_lo, w_hi = {"c3": (3072, 768, 24, 128, 5, 30), "c5": (1024, 1024, 8, 64, 4, 12),
                                          "anet": (1024, 1024, 16, 128, 5, 30)}[config]
        for i in range(n):
            L = int(rs.randint(l_lo, l_hi + 1))
            v = rs.standard_normal((L, dv)).astype(np.float32)
            v /= np.linalg.norm(v, axis=1, keepdims=True)
            tv = rs.standard_normal((L, 512)).astype(np.float32)
            if config == "c3":
                nc = 5
            elif config == "c5":
                nc = int(rs.choice([1, 2, 3, 4, 5, 6], p=[0.30, 0.35, 0.20, 0.08, 0.05, 0.02]))
            else:       # ActivityNet Captions: 37,421 / 10,009 = 3.7 per video, 1..10
                nc = int(rs.choice(np.arange(1, 11), p=[0.05, 0.17, 0.27, 0.22, 0.13, 0.07, 0.04, 0.03, 0.01, 0.01]))
            caps = [rs.standard_normal((int(rs.randint(w_lo, w_hi + 1)), dq)).astype(np.float32) for _ in range(nc)]
            tcaps = [rs.standard_normal((1, 512)).astype(np.float32) for _ in range(nc)]
            self.items.append((v, caps, tv, tcaps, i, [f"v{i}#{c}" for c in range(nc)], f"v{i}"))

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        it = self.items[i % len(self.items)]
        return it if i < len(self.items) else it[:4] + (i, [f"v{i}#{c}" for c in range(len(it[1]))], f"v{i}")


def run(n_videos=1024, workers=4, prec="bf16", dev="cuda:0"):
    from dldkd_amd import ops, train as T
    from dldkd_amd.model import DLDKD
    cfg = types.SimpleNamespace(visual_input_size=3072, query_input_size=768, inheritance_hidden=384, exploration_hidden=384,
                                max_ctx_l=128, max_desc_l=30, input_drop=0.2, drop=0.2, n_heads=4, initializer_range=0.02,
                                margin=0.1, use_hard_negative=True, hard_pool_size=20, label_style="soft")
    ds = SynthTrainSet(n_videos)
    out = {"n_videos": n_videos, "batch": 128, "precision": prec, "loader_workers": workers}
    ops.set_gemm_precision(prec)
    try:
        for name, resident in (("host_loader", False), ("device_resident", True)):
            opt = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                        collection="tvr", alpha=0.8, belta=0.8, device=torch.device(dev), bsz=128, pin_memory=True,
                                        num_workers=workers, lr=3e-4, wd=0.01, lr_warmup_proportion=0.01, n_epoch=100,
                                        hard_negative_start_epoch=0, hard_pool_size=20, distill_loss_decay="exp", exponential_k=0.95,
                                        selfDistil_sigmoid_k=800, alpha_decay="sigmoid", belta_decay="sigmoid", grad_clip=-1,
                                        device_resident_train=resident)
            torch.manual_seed(0)
            m = DLDKD(cfg, opt).to(dev)
            t0 = time.perf_counter()
            loader = T.make_train_loader(ds, opt, 0, 1)
            torch.cuda.synchronize()
            build = time.perf_counter() - t0
            optim = T.make_optimizer(m, opt, len(loader))
            stepper = T.GraphedTrainStep(m, optim, opt, defer_loss_float=True)
            T.train_epoch(m, loader, optim, opt, 0, stepper=stepper)          # warm-up epoch (captures)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for ep in (1, 2):
                T.train_epoch(m, loader, optim, opt, ep, stepper=stepper)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[name] = {"steps_per_s": 2 * len(loader) / dt, "ms_per_step_wall": dt / (2 * len(loader)) * 1e3,
                         "setup_s": build, "replays": stepper.replays, "captures": stepper.captures}
    finally:
        ops.set_gemm_precision("fp32")
    return out


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 4)))
