"""Where the wall-clock of train.train_epoch goes with the device-resident data path (opt.device_resident_train) at TVR shapes:
  python3 tools/prof_train_epoch.py [n_videos=4096] [precision=bf16] [config=c3|c5] [--profile] [--no-prefetch]
Prints one JSON object: per-step wall inside an epoch (steady state: host enqueue time per step and the epoch's GPU-inclusive
wall / steps), the per-epoch fixed cost (schedule switch, loss read-back), captures / replays / eager steps, and with --profile the
cProfile top of one epoch.  (The step alone, inputs resident and the loss deferred: bench.py extras c3_train_step_ms_*_deferred_loss.)"""
import cProfile, io, json, os, pstats, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("dl-dkd_amd", "tests/golden", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch


def run(n_videos=4096, prec="bf16", profile=False, dev="cuda:0", epochs=3, prefetch=True, ds=None, config="c3"):
    from bench_train_loader import SynthTrainSet
    from dldkd_amd import ops, train as T
    from dldkd_amd.model import DLDKD
    dv, dq, lmax, drop = {"c3": (3072, 768, 128, 0.2), "c5": (1024, 1024, 64, 0.15), "anet": (1024, 1024, 128, 0.2)}[config]
    cfg = types.SimpleNamespace(visual_input_size=dv, query_input_size=dq, inheritance_hidden=384, exploration_hidden=384,
                                max_ctx_l=lmax, max_desc_l=30, input_drop=drop, drop=drop, n_heads=4, initializer_range=0.02,
                                margin=0.1, use_hard_negative=True, hard_pool_size=20, label_style="soft")
    opt = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                collection="tvr", alpha=0.8, belta=0.8, device=torch.device(dev), bsz=128, pin_memory=True,
                                num_workers=0, lr=3e-4, wd=0.01, lr_warmup_proportion=0.01, n_epoch=100,
                                hard_negative_start_epoch=0, hard_pool_size=20, distill_loss_decay="exp", exponential_k=0.95,
                                selfDistil_sigmoid_k=800, alpha_decay="sigmoid", belta_decay="sigmoid", grad_clip=-1,
                                device_resident_train=True, prefetch_batches=prefetch)
    ds = ds if ds is not None else SynthTrainSet(n_videos, config=config)
    out = {"config": config, "n_videos": n_videos, "batch": 128, "precision": prec, "prefetch_batches": prefetch,
           "query_bucket": T.GraphedTrainStep.QUERY_BUCKET}
    ops.set_gemm_precision(prec)
    try:
        torch.manual_seed(0)
        m = DLDKD(cfg, opt).to(dev)
        t0 = time.perf_counter()
        loader = T.make_train_loader(ds, opt, 0, 1)
        torch.cuda.synchronize()
        out["setup_s"] = time.perf_counter() - t0
        optim = T.make_optimizer(m, opt, len(loader))
        stepper = T.GraphedTrainStep(m, optim, opt, defer_loss_float=True)
        T.train_epoch(m, loader, optim, opt, 0, stepper=stepper)              # warm-up epoch (first sight + capture)
        torch.cuda.synchronize()
        walls = []
        for ep in range(1, 1 + epochs):
            t0 = time.perf_counter()
            T.train_epoch(m, loader, optim, opt, ep, stepper=stepper)
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)
        steps = len(loader)
        out["queries_per_batch"] = sorted({len(p.labels) for p in loader.plans()}) if hasattr(loader, "plans") else None
        out["max_memory_allocated_gb"] = torch.cuda.max_memory_allocated() / 1e9
        out["memory_reserved_gb"] = torch.cuda.memory_reserved() / 1e9
        out.update(steps_per_epoch=steps, epoch_wall_s=walls, ms_per_step_wall=[w / steps * 1e3 for w in walls],
                   captures=stepper.captures, replays=stepper.replays, eager_steps=stepper.eager_steps,
                   prefetched=stepper.prefetched, fallbacks=[list(f) for f in stepper.fallbacks])
        # host enqueue time of the loop body alone (no synchronisation): what the host needs per step
        it = iter(loader)
        batches = [next(it) for _ in range(min(8, steps))]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches:
            stepper(b)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        out["host_enqueue_ms_per_step_given_batches"] = host / len(batches) * 1e3
        t0 = time.perf_counter()
        it = iter(loader)
        for _ in range(len(batches)):
            next(it)
        out["loader_host_ms_per_batch"] = (time.perf_counter() - t0) / len(batches) * 1e3
        torch.cuda.synchronize()
        if profile:
            pr = cProfile.Profile()
            pr.enable()
            T.train_epoch(m, loader, optim, opt, 1 + epochs, stepper=stepper)
            torch.cuda.synchronize()
            pr.disable()
            s = io.StringIO()
            pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
            out["cprofile"] = s.getvalue().splitlines()[:60]
    finally:
        ops.set_gemm_precision("fp32")
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    r = run(int(args[0]) if args else 4096, args[1] if len(args) > 1 else "bf16", "--profile" in sys.argv,
            prefetch="--no-prefetch" not in sys.argv, config=args[2] if len(args) > 2 else "c3")
    prof = r.pop("cprofile", None)
    print(json.dumps(r))
    if prof:
        print("\n".join(prof))
