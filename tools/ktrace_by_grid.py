"""Aggregate a rocprofv3 kernel trace by (kernel name, grid size): which launch shapes cost the time."""
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(fs[0])):
    if pat not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"][:60], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg[key][0] += 1
    agg[key][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k[0]:60s} grid {k[1]:>6s} {k[2]:>5s} {k[3]:>5s}  calls {n:5d}  total {t/1e6:9.3f} ms  avg {t/n/1e3:8.1f} us")
