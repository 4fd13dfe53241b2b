"""Wall-clock of train.train() itself - the reference's driver loop (method/train.py:191-247): epochs of train_epoch, an eval_epoch on
the validation sets after every epoch, the best checkpoint - on in-memory TVR-shaped data:
  python3 tools/prof_train_run.py [n_train=2048] [n_val=1089] [precision=bf16] [epochs=6] [pool] [--profile]
Prints the run's wall per epoch (training part / evaluation part, from the log's timestamps taken around the two calls), the
one-time costs in front (device tables, captures), and with --profile the cProfile top of the whole run."""
import cProfile, io, json, os, pstats, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("dl-dkd_amd", "tests/golden", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import torch


class _L(torch.utils.data.Dataset):
    def __init__(self, x): self.x = x
    def __len__(self): return len(self.x)
    def __getitem__(self, i): return self.x[i]


def val_sets(n, seed=1, dv=3072, dq=768):
    """VisDataSet4DLDKD / TxtDataSet4DLDKD stand-ins (data_provider.py:307-309, 344-354): (feat, index, id) items."""
    rs = np.random.RandomState(seed)
    vids, txts = [], []
    for i in range(n):
        L = int(rs.randint(24, 129))
        v = rs.standard_normal((L, dv)).astype(np.float32)
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        vids.append((torch.from_numpy(v), i, f"v{i}"))
        for c in range(5):
            w = rs.standard_normal((int(rs.randint(5, 31)), dq)).astype(np.float32)
            txts.append((torch.from_numpy(w), len(txts), f"v{i}#enc#{c}"))
    return _L(vids), _L(txts)


def run(n_train=2048, n_val=1089, prec="bf16", epochs=6, profile=False, dev="cuda:0", pool=None):
    from bench_train_loader import SynthTrainSet
    from dldkd_amd import train as T
    from dldkd_amd import eval as E
    from dldkd_amd.model import DLDKD
    cfg = types.SimpleNamespace(visual_input_size=3072, query_input_size=768, inheritance_hidden=384, exploration_hidden=384,
                                max_ctx_l=128, max_desc_l=30, input_drop=0.2, drop=0.2, n_heads=4, initializer_range=0.02,
                                margin=0.1, use_hard_negative=False, hard_pool_size=20, label_style="soft")
    opt = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                collection="tvr", alpha=0.8, belta=0.8, device=torch.device(dev), bsz=128, pin_memory=True,
                                num_workers=0, lr=3e-4, wd=0.01, lr_warmup_proportion=0.01, n_epoch=epochs, max_es_cnt=-1,
                                hard_negative_start_epoch=2, hard_pool_size=20, distill_loss_decay="exp", exponential_k=0.95,
                                selfDistil_sigmoid_k=800, alpha_decay="sigmoid", belta_decay="sigmoid", grad_clip=-1,
                                eval_context_bsz=200, eval_query_bsz=50, train_precision=prec,
                                ckpt_filepath="/tmp/prof_train_run.ckpt")
    ds = SynthTrainSet(n_train, pool=pool)
    vv, vt = val_sets(n_val)
    marks = []
    orig_te, orig_ee = T.train_epoch, T.eval_epoch

    def te(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = orig_te(*a, **k)
        torch.cuda.synchronize(); marks.append(("train", time.perf_counter() - t0))
        return r

    def ee(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = orig_ee(*a, **k)
        torch.cuda.synchronize(); marks.append(("eval", time.perf_counter() - t0))
        return r
    T.train_epoch, T.eval_epoch = te, ee
    torch.manual_seed(0)
    m = DLDKD(cfg, opt)
    pr = cProfile.Profile() if profile else None
    try:
        t0 = time.perf_counter()
        if pr:
            pr.enable()
        hist = T.train(m, ds, vv, vt, opt)
        torch.cuda.synchronize()
        if pr:
            pr.disable()
        total = time.perf_counter() - t0
    finally:
        T.train_epoch, T.eval_epoch = orig_te, orig_ee
        E.clear_feature_cache()
    steps = -(-n_train // 128)
    tr = [t for k, t in marks if k == "train"]
    ev = [t for k, t in marks if k == "eval"]
    out = {"n_train": n_train, "n_val_videos": n_val, "n_val_captions": 5 * n_val, "precision": prec, "epochs": len(tr), "steps_per_epoch": steps,
           "total_s": total, "train_epoch_s": tr if len(tr) <= 12 else tr[:6] + ["..."] + tr[-6:], "eval_epoch_s": ev if len(ev) <= 12 else ev[:6] + ["..."] + ev[-6:],
           "train_s_sum": sum(tr), "eval_s_sum": sum(ev), "train_ms_per_step_median": sorted(t / steps * 1e3 for t in tr)[len(tr) // 2],
           "outside_train_and_eval_s": total - sum(tr) - sum(ev), "sumr": [h[2] for h in hist]}
    if pr:
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30)
        out["cprofile"] = s.getvalue().splitlines()[:64]
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    r = run(int(args[0]) if args else 2048, int(args[1]) if len(args) > 1 else 1089, args[2] if len(args) > 2 else "bf16",
            int(args[3]) if len(args) > 3 else 6, "--profile" in sys.argv, pool=int(args[4]) if len(args) > 4 else None)
    prof = r.pop("cprofile", None)
    print(json.dumps(r))
    if prof:
        print("\n".join(prof))
