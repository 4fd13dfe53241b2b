"""Where do the gradient collectives of the data-parallel step sit in time?  One-rank RCCL group forced on (the box has one GPU),
C3-size batch, train.GraphedTrainStep with one gradient bucket per tower: the step replays as 4 graph segments + the optimizer
graph, each bucket's all-reduce + 1/world scaling issued on the comm stream behind its segment.

    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/prof_ddp_overlap.py run
    python3 tools/prof_ddp_overlap.py report <dir> [out.json]

`run` also times the plain (single-graph, no collective) stepper and the bucketed one with HIP events.  `report` reads the
kernel trace: per step, the comm-stream kernels (everything not on the step's own queue between two optimizer updates) with
their offsets from the step's first kernel, the end of the backward pass, and how much of the comm work lies after it.
(RCCL launches no kernel for an in-place all-reduce over ONE rank, so on this box the bucket's scaling kernel marks where its
collective executes; with peers the all-reduce kernel sits at the same point of the comm stream.)"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in ("dl-dkd_amd", "tests/golden", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p_))


def run():
    import torch
    import torch.distributed as dist
    import bench_train
    from dldkd_amd import ops
    from dldkd_amd import train as T
    from dldkd_amd.optimization import BertAdam
    dev = "cuda:0"
    topt = __import__("types").SimpleNamespace(grad_clip=-1)
    out = {}
    for prec in ("bf16",):
        ops.set_gemm_precision(prec)
        m, opt, batch = bench_train.build("c3", 0.2, dev)
        plain = T.GraphedTrainStep(m, opt, topt, defer_loss_float=True)
        out[f"plain_{prec}"] = bench_train.timed(lambda: plain(batch), 30, 10)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29597", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device(dev))
        T.DDP_MIN_WORLD = 1
        try:
            for name, buckets in (("ddp_one_bucket", False), ("ddp_tower_buckets", True)):
                m, _, batch = bench_train.build("c3", 0.2, dev)
                opt = BertAdam([{"params": list(m.parameters()), "weight_decay": 0.01}], lr=3e-4, warmup=0.01, t_total=100000,
                               grad_buckets=m.grad_buckets() if buckets else None)
                st = T.GraphedTrainStep(m, opt, topt, defer_loss_float=True)
                out[f"{name}_{prec}"] = bench_train.timed(lambda: st(batch), 30, 10)
                if buckets:
                    e = next(iter(st.graphs.values()))
                    out["segments"] = len(e.segments)
                    out["bucket_MB"] = [round(4 * (hi - lo) / 1e6, 2) for lo, hi in opt.fp.bucket_ranges]
        finally:
            T.DDP_MIN_WORLD = 2
            dist.destroy_process_group()
        ops.set_gemm_precision("fp32")
    print(json.dumps(out))


def report(d, out_path=None):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"),
                         r.get("Stream_Id", "0")))
    rows.sort()
    upd = [i for i, r in enumerate(rows) if "adam_update_kernel" in r[2]]
    # the last 30 steps belong to the bucketed stepper; take the last few
    steps = []
    for a, b in zip(upd[-6:-1], upd[-5:]):
        seg = rows[a + 1:b + 1]
        queues = {}
        for r in seg:
            queues.setdefault((r[3], r[4]), []).append(r)
        main = max(queues.values(), key=len)
        t0 = main[0][0]
        comm = [r for k, v in queues.items() if v is not main for r in v]
        bwd_end = max(r[1] for r in main if "adam" not in r[2] and "zero_f32" not in r[2])
        steps.append({
            "step_us": (main[-1][1] - t0) / 1e3, "main_kernels": len(main), "backward_end_us": (bwd_end - t0) / 1e3,
            "comm": [{"kernel": r[2][:60], "start_us": (r[0] - t0) / 1e3, "dur_us": (r[1] - r[0]) / 1e3} for r in comm],
            "comm_us_after_backward_end": sum(max(0, r[1] - max(r[0], bwd_end)) for r in comm) / 1e3})
    res = {"steps": steps}
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
