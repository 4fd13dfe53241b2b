"""Experiment (GPU box): does hipStreamWaitValue32 park a stream on a counter that a kernel on another stream bumps?
Run under `timeout`: a wait that never returns must not take the box down."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dl-dkd_amd"))
import torch
from dldkd_amd import native

dev = torch.device("cuda:0")
L = native.lib()
side = torch.cuda.Stream()
done = torch.zeros(4, dtype=torch.int32, device=dev)
out = torch.zeros(1 << 20, device=dev)
torch.cuda.synchronize()

# 1. already-satisfied wait (cannot hang)
done.fill_(5)
torch.cuda.synchronize()
with torch.cuda.stream(side):
    native.check(L.dldkd_stream_wait_counter(native.stream(), native.ptr(done), 3), "wait")
    out.fill_(1.0)
side.synchronize()
print("satisfied wait ok", float(out[0]))

# 2. wait released by a later kernel on the main stream
done.zero_()
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(side):
    ev0.record()
    native.check(L.dldkd_stream_wait_counter(native.stream(), native.ptr(done), 7), "wait")
    out.fill_(2.0)
    ev1.record()
time.sleep(0.2)
print("side stream done before release? ", ev1.query())
torch.cuda._sleep(int(2e8))          # ~0.1 s of spinning on the main stream
done.fill_(7)
t0 = time.time()
while not ev1.query() and time.time() - t0 < 10:
    time.sleep(0.01)
print("released:", ev1.query(), "out", float(out[0]), "waited ms", ev0.elapsed_time(ev1) if ev1.query() else None)

# 3. the scorer's own counters: side stream finishes range 0 while later ranges still run
from dldkd_amd import scoring
g = torch.Generator(device=dev).manual_seed(4)
nv, nq = 615, 17505
gal = [torch.randn(nv, 128, 384, generator=g, device=dev) for _ in range(2)]
qs = [torch.randn(nq, 384, generator=g, device=dev) for _ in range(2)]
pg = scoring.pack_gallery(gal, None)
pq = scoring.pack_queries(qs)
n, per = scoring.plan_query_split(nq, nv, 2, min_split=4)
ws = scoring.simpool_partials(pq, pg, q_split=1)
ref = scoring.simpool_finish(ws, pq, pg)[0]
torch.cuda.synchronize()
wg0 = (nv + 3) // 4 * 2
cnt = torch.zeros(n, dtype=torch.int32, device=dev)
blocks = [torch.empty(min(per, nq - r * per), nv, device=dev) for r in range(n)]
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 2)]
main = torch.cuda.current_stream()
for it in range(3):
    cnt.zero_()
    e0 = torch.cuda.Event()
    e0.record(main)
    evs[0].record(main)
    ws2 = scoring.simpool_partials(pq, pg, ws, q_split=n, done=cnt)
    evs[1].record(main)
    with torch.cuda.stream(side):
        side.wait_event(e0)
        for r in range(n):
            native.check(L.dldkd_stream_wait_counter(native.stream(), native.ptr(cnt[r:]), wg0), "wait")
            scoring.simpool_finish(ws2, pq, pg, q_range=(r * per, min((r + 1) * per, nq)), out=blocks[r])
            evs[2 + r].record()
    main.wait_stream(side)
    torch.cuda.synchronize()
    print("iter", it, "scorer ms", evs[0].elapsed_time(evs[1]), "range finish times ms", [round(evs[0].elapsed_time(evs[2 + r]), 3) for r in range(n)],
          "equal", torch.equal(torch.cat(blocks, 0), ref))
