"""Time the training step on the GPU box: C3 (TVR: 128 videos / 640 queries, soft labels, hard negatives) or C5 (Charades
rank-local: 128 videos / 257 queries, 1024-d), eager and hipGraph-replayed (train.GraphedTrainStep).

    python tools/bench_train.py [--config c3|c5] [--prec fp32|bf16] [--drop 0.2] [--steps 30] [--warmup 10]

Per mode: HIP-event time of every step (stream time) AND host wall time, median / p90 over `steps` after `warmup`."""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("dl-dkd_amd", "tests/golden"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch  # noqa: E402
import synth  # noqa: E402


def build(config, drop, dev):
    from dldkd_amd.model import DLDKD
    from dldkd_amd.optimization import BertAdam
    dv, dq = (3072, 768) if config == "c3" else (1024, 1024)
    cfg = types.SimpleNamespace(visual_input_size=dv, query_input_size=dq, inheritance_hidden=384, exploration_hidden=384,
                                max_ctx_l=128, max_desc_l=30, input_drop=drop, drop=drop, n_heads=4, initializer_range=0.02,
                                margin=0.1, use_hard_negative=True, hard_pool_size=20, label_style="soft")
    opt_ = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                 collection="tvr", alpha=0.8, belta=0.8)
    torch.manual_seed(0)
    m = DLDKD(cfg, opt_).to(dev).train()
    opt = BertAdam([{"params": list(m.parameters()), "weight_decay": 0.01}], lr=3e-4, warmup=0.01, t_total=100000)
    if config == "c3":
        batch = synth.make_train_batch(3, nv=128, caps=5, L=128, len_lo=24, dv=dv, dq=dq)
    else:
        batch = synth.make_train_batch(5, nv=128, caps=sorted([3] + [2] * 127, reverse=True), L=64, len_lo=8, dv=dv, dq=dq)
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    return m, opt, batch


def timed(step, steps, warmup):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    walls = []
    ev[0].record()
    t_all = time.perf_counter()
    for i in range(steps):
        t0 = time.perf_counter()
        step()
        ev[i + 1].record()
        walls.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t_all) * 1e3 / steps
    gpu = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    walls.sort()
    q = lambda a, f: a[min(len(a) - 1, int(f * len(a)))]      # noqa: E731
    return {"ms_per_step_wall_mean": total, "stream_ms_median": q(gpu, 0.5), "stream_ms_p90": q(gpu, 0.9),
            "host_ms_median": q(walls, 0.5), "host_ms_p90": q(walls, 0.9)}


def run(config="c3", prec="bf16", drop=0.2, steps=30, warmup=10, dev="cuda:0", modes=("eager", "graph"), tower_streams=None):
    from dldkd_amd import ops
    from dldkd_amd import train as T
    ops.set_gemm_precision(prec)
    out = {"config": config, "precision": prec, "dropout": drop, "steps": steps, "warmup": warmup}
    try:
        topt = types.SimpleNamespace(grad_clip=-1)
        if "eager" in modes:
            m, opt, batch = build(config, drop, dev)
            if tower_streams is not None:
                m.tower_streams = tower_streams
            out["eager"] = timed(lambda: T.train_step(m, batch, opt, topt), steps, warmup)
        if "graph" in modes:
            m, opt, batch = build(config, drop, dev)
            if tower_streams is not None:
                m.tower_streams = tower_streams
            g = T.GraphedTrainStep(m, opt, topt, defer_loss_float=False)
            out["graph"] = timed(lambda: g(batch), steps, warmup)
            out["graph"]["replays"], out["graph"]["eager_steps"], out["graph"]["captures"] = g.replays, g.eager_steps, g.captures
            g2 = T.GraphedTrainStep(m, opt, topt, defer_loss_float=True)
            out["graph_no_loss_sync"] = timed(lambda: g2(batch), steps, warmup)
            # the same replayed step with the batch ALREADY in the stepper's input buffers - what a data path that fills them itself
            # gives (train.py device_resident_train: the batch is gathered into them by a kernel; GraphedTrainStep._replay copies
            # only tensors that live elsewhere): no 201-MB device-to-device copy at the head of the C3 step
            e = next(reversed(g2.graphs.values()), None) if getattr(g2, "graphs", None) else None
            if e is not None and hasattr(e, "static"):
                sb = dict(batch)
                for k in g2.TENSOR_KEYS:
                    if e.static[k].shape == batch[k].shape:
                        e.static[k].copy_(batch[k])
                        sb[k] = e.static[k]
                torch.cuda.synchronize()
                out["graph_static_inputs"] = timed(lambda: g2(sb), steps, warmup)
            # double-buffered inputs: the caller hands over the NEXT batch with the current one (train_epoch does, through
            # GraphedTrainStep.iterate); it is copied into the other buffer set while this step runs.  Two DIFFERENT batches take
            # turns, every copy is real; float(loss) every step / deferred
            m, opt, batch = build(config, drop, dev)
            if tower_streams is not None:
                m.tower_streams = tower_streams
            other = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
            for k in ("student_videos", "teacher_videos", "student_text", "teacher_text"):
                other[k] = other[k].flip(0) if k.endswith("videos") else other[k] * 0.5
            pair = [batch, other]
            for name, defer in (("graph_prefetch", False), ("graph_prefetch_no_loss_sync", True)):
                gp = T.GraphedTrainStep(m, opt, topt, defer_loss_float=defer)
                turn = [0]

                def step():
                    i = turn[0]
                    turn[0] = 1 - i
                    return gp(pair[i], next_batch=pair[1 - i])
                out[name] = timed(step, steps, warmup)
                out[name].update(replays=gp.replays, eager_steps=gp.eager_steps, captures=gp.captures, prefetched=gp.prefetched)
    finally:
        ops.set_gemm_precision("fp32")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--prec", default="bf16")
    ap.add_argument("--drop", type=float, default=0.2)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--modes", default="eager,graph")
    ap.add_argument("--tower-streams", type=int, default=-1, help="1 / 0: the four towers on four streams / one; -1: the default")
    ap.add_argument("--fused-towers", type=int, default=1, help="0: the unfused kernel chain of the training towers (A/B)")
    ap.add_argument("--fused-losses", type=int, default=1, help="0: one kernel pair per loss term (A/B)")
    a = ap.parse_args()
    from dldkd_amd import functional as F_
    F_.TOWER_TRAIN_FUSED, F_.BRANCH_LOSS_FUSED = bool(a.fused_towers), bool(a.fused_losses)
    print(json.dumps(run(a.config, a.prec, a.drop, a.steps, a.warmup, modes=tuple(a.modes.split(",")),
                         tower_streams=None if a.tower_streams < 0 else bool(a.tower_streams)), indent=1))
