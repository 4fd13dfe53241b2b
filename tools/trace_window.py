"""Kernel timeline of the last N launches of a kernel (and everything between them) from a rocprofv3 kernel trace CSV:
python3 tools/trace_window.py <dir> <kernel substring> [N]   - name, start offset (us), duration (us), gap to the previous end."""
import csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 22
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if pat in r[2]]
lo = idx[-n] if len(idx) >= n else idx[0]
t0, prev = rows[lo][0], rows[lo][0]
for s, e, name in rows[max(lo - 3, 0):idx[-1] + 2]:
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev) / 1e3:8.1f}  {name[:70]}")
    prev = e
