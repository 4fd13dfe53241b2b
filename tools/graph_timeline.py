"""Graph-level timeline of the replayed training step WITHOUT a profiler (python3 tools/graph_timeline.py [c3|c5] [sync|defer]):
timing events around every graph of GraphedTrainStep._replay_parallel, on the graph's own stream; start / end of each graph
relative to the step's first event, median over the steps.  (Under rocprofv3 a hipGraphLaunch costs 30 - 160 us on the host - the
tracer instruments every packet - and the profiled timeline shows launch-order artefacts that the real run does not have.)"""
import json
import statistics
import sys
import types

sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tools")
import bench_train as B  # noqa: E402
import torch  # noqa: E402


def main(config="c5", mode="sync", steps=30, warmup=10, prec="bf16"):
    from dldkd_amd import ops
    from dldkd_amd import train as T
    ops.set_gemm_precision(prec)
    m, opt, batch = B.build(config, 0.2, "cuda:0")
    g = T.GraphedTrainStep(m, opt, types.SimpleNamespace(grad_clip=-1), defer_loss_float=(mode != "sync"))
    for _ in range(warmup):
        g(batch)
    rec = []

    def E():
        return torch.cuda.Event(enable_timing=True)

    def replay_parallel(e):
        par, main = e.par, g.stream
        streams, ev = par["streams"], par["ev"]
        marks = {}

        def span(name, stream, graph):
            a, b = E(), E()
            a.record(stream)
            graph.replay()
            b.record(stream)
            marks[name] = (a, b)

        t0 = E()
        t0.record(main)
        par["ev_pre"].record(main)
        pre0 = par.get("pre0")
        if pre0 is not None:                                # (round 6: both video towers' input LayerNorm, on the first video tower's stream)
            s0 = streams[par["pre0_tower"]]
            s0.wait_event(par["ev_in_video"] if g.EARLY_VIDEO_START else par["ev_pre"])
            with torch.cuda.stream(s0):
                span("pre0_ln_dual", s0, pre0)
                par["ev_ln"].record(s0)
        span("pre", main, par["pre"])
        par["ev_pre_done"].record(main)
        for i in par["order"]:
            early = g.EARLY_VIDEO_START and par["video"][i] and streams[i] is not main
            if streams[i] is not main:
                if pre0 is not None and par["video"][i]:
                    if i != par["pre0_tower"]:
                        streams[i].wait_event(par["ev_ln"])
                else:
                    streams[i].wait_event(par["ev_in_video"] if early else par["ev_pre"])
            with torch.cuda.stream(streams[i]):
                span(f"fwd{i}{'v' if par['video'][i] else 'q'}", streams[i], par["fwd"][i])
                ev["fwd"][i].record(streams[i])
        for b, (gr, si, towers) in enumerate(par["loss"]):
            if streams[si] is not main:
                streams[si].wait_event(par["ev_pre_done"])
            for t in towers:
                if streams[t] is not streams[si]:
                    streams[si].wait_event(ev["fwd"][t])
            with torch.cuda.stream(streams[si]):
                span(f"loss{b}", streams[si], gr)
                ev["loss"][b].record(streams[si])
        for i in par["order"]:
            if streams[par["loss"][par["loss_of"][i]][1]] is not streams[i]:
                streams[i].wait_event(ev["loss"][par["loss_of"][i]])
            with torch.cuda.stream(streams[i]):
                span(f"bwd{i}{'v' if par['video'][i] else 'q'}", streams[i], par["bwd"][i])
                ev["bwd"][i].record(streams[i])
        for i, x in enumerate(ev["bwd"]):
            if streams[i] is not main:
                main.wait_event(x)
        if par["tail"] is not None:
            span("tail", main, par["tail"])
        span("opt", main, par["opt"])
        rec.append((t0, marks))

    g._replay_parallel = replay_parallel
    torch.cuda.synchronize()
    for _ in range(steps):
        g(batch)
    torch.cuda.synchronize()
    out = {}
    for t0, marks in rec:
        for k, (a, b) in marks.items():
            out.setdefault(k, []).append((t0.elapsed_time(a) * 1e3, t0.elapsed_time(b) * 1e3))
    res = {k: (round(statistics.median(x[0] for x in v), 1), round(statistics.median(x[1] for x in v), 1)) for k, v in out.items()}
    print(json.dumps({"config": config, "mode": mode, "graphs_start_end_us": res}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "c5", sys.argv[2] if len(sys.argv) > 2 else "sync",
         prec=sys.argv[3] if len(sys.argv) > 3 else "bf16")
