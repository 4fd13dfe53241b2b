"""Timeline of ONE replayed training step from a rocprofv3 kernel trace (python3 tools/step_timeline.py <dir> [min_us]): every
kernel between the last two optimizer updates with its start offset, duration, queue, and how many kernels run beside it; then
the intervals during which only one queue is busy (the serial sections of the step)."""
import csv, glob, sys
d = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
upd = [i for i, r in enumerate(rows) if "adam_update_kernel" in r[2]]
a, b = upd[-2], upd[-1]
seg = rows[a + 1:b + 1]
t0 = seg[0][0]
print(f"step: {(seg[-1][1] - t0) / 1e3:.1f} us, {len(seg)} kernels, kernel time sum {sum(e - s for s, e, _, _ in seg) / 1e3:.1f} us")
for s, e, name, q in seg:
    if (e - s) / 1e3 >= min_us:
        conc = sum(1 for s2, e2, _, _ in seg if s2 < e and e2 > s) - 1
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  q{q:>3s}  beside {conc:2d}  {name[:80]}")
