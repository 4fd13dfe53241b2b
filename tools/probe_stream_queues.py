import torch, time
dev = torch.device("cuda:0")
torch.cuda.init()
cands = [torch.cuda.Stream(device=dev) for _ in range(12)]
cyc = 2_000_000
def run(streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in streams:
        with torch.cuda.stream(s):
            torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
for _ in range(2): run(cands[:1])
one = min(run([cands[0]]) for _ in range(3))
print("one", one)
n = len(cands)
for i in range(n):
    row = []
    for j in range(n):
        row.append("%.1f" % (min(run([cands[i], cands[j]]) for _ in range(2)) / one) if i != j else " - ")
    print(i, " ".join(row))
print("all 4 first:", run(cands[:4]) / one, " streams 0,1,2,3")
