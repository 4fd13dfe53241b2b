"""Host cost of replaying the captured training step: wall time of the enqueue calls alone (no synchronisation inside the timed
region), single graph vs the parallel tower graphs.  python tools/prof_replay_host.py [c3] [bf16]"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "dl-dkd_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tools")]
import torch
import bench_train as B
from dldkd_amd import ops, train as T

cfgname, prec = (sys.argv + ["c3", "bf16"])[1:3]
ops.set_gemm_precision(prec)
for par in (False, True):
    m, opt, batch = B.build(cfgname, 0.2, "cuda:0")
    topt = types.SimpleNamespace(grad_clip=-1, parallel_tower_graphs=par)
    g = T.GraphedTrainStep(m, opt, topt, defer_loss_float=True)
    for _ in range(6):
        g(batch)
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g(batch)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
    host.sort(); total.sort()
    print(f"parallel_tower_graphs={par}: host enqueue {host[10]:.2f} ms, enqueue + drain {total[10]:.2f} ms  (replays {g.replays})")
