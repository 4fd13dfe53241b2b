"""Do hipGraph launches on different streams run side by side?  T graphs of N one-workgroup spin kernels each, every graph on
its own stream (streams measured to have their own hardware queues), launched back to back behind one event."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dl-dkd_amd"))
import torch
from dldkd_amd.staging import concurrent_streams
dev = torch.device("cuda:0")
T_, N, CYC = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 40000
streams = concurrent_streams(dev, T_ + 1)
main, sides = streams[0], streams[1:]
graphs = []
for s in sides:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(N):
            torch.cuda._sleep(CYC)
    graphs.append(g)
def eager_one():
    with torch.cuda.stream(sides[0]):
        for _ in range(N):
            torch.cuda._sleep(CYC)
def run(k, graph=True):
    torch.cuda.synchronize()
    ev = torch.cuda.Event()
    t0 = time.perf_counter()
    with torch.cuda.stream(main):
        torch.cuda._sleep(CYC * 20)          # the towers wait for work on the main stream, as the loss graph
        ev.record(main)
    for s, g in list(zip(sides, graphs))[:k]:
        s.wait_event(ev)
        with torch.cuda.stream(s):
            if graph:
                g.replay()
            else:
                for _ in range(N):
                    torch.cuda._sleep(CYC)
    th = time.perf_counter()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, (th - t0) * 1e3
for graph in (True, False):
    for k in (1, 2, 4):
        run(k, graph)
        r = sorted(run(k, graph) for _ in range(5))[2]
        print(f"{'graphs' if graph else 'eager '} x{k}: total {r[0]:.2f} ms (host enqueue {r[1]:.2f} ms)   [{N} kernels of ~{CYC} cycles each per stream]")
