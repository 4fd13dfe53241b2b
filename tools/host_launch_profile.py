"""Where the HOST's time goes in one replayed training step (python3 tools/host_launch_profile.py [c3|c5]): wall time of every
graph launch / copy / host preparation of GraphedTrainStep._replay, median over the steps, with float(loss) every step (the GPU is
idle when a step begins, so these are launch costs, not back-pressure).  Compared with the step's stream time."""
import json
import statistics
import sys
import time
import types

sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tools")
import bench_train as B  # noqa: E402
import torch  # noqa: E402


def main(config="c5", steps=40, warmup=10):
    from dldkd_amd import ops
    from dldkd_amd import train as T
    ops.set_gemm_precision("bf16")
    m, opt, batch = B.build(config, 0.2, "cuda:0")
    g = T.GraphedTrainStep(m, opt, types.SimpleNamespace(grad_clip=-1), defer_loss_float=False)
    marks = {}

    def wrap(obj, name, label):
        fn = getattr(obj, name)

        def timed(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                marks.setdefault(label, []).append((time.perf_counter() - t0) * 1e6)
        setattr(obj, name, timed)

    for _ in range(warmup):
        g(batch)
    e = next(iter(g.graphs.values()))
    par = e.par
    for i, gr in enumerate(par["fwd"]):
        wrap(gr, "replay", f"fwd{i}")
    for i, gr in enumerate(par["bwd"]):
        wrap(gr, "replay", f"bwd{i}")
    for i, (gr, _, _) in enumerate(par["loss"]):
        wrap(gr, "replay", f"loss{i}")
    for k in ("pre", "tail", "opt"):
        wrap(par[k], "replay", k)
    wrap(g, "_replay_parallel", "replay_parallel(all graphs)")
    wrap(g, "_replay", "_replay(total host)")
    wrap(opt, "host_prepare", "host_prepare")
    wrap(m, "_draw_triplet", "draw_triplet")
    torch.cuda.synchronize()
    walls, ev = [], []
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        t0 = time.perf_counter()
        g(batch)
        walls.append((time.perf_counter() - t0) * 1e6)
        b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    out = {k: round(statistics.median(v), 1) for k, v in marks.items()}
    out["step_wall_us"] = round(statistics.median(walls), 1)
    out["step_stream_us"] = round(statistics.median(a.elapsed_time(b) for a, b in ev) * 1e3, 1)
    print(json.dumps({"config": config, **out}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "c5")
