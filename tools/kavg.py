"""python3 tools/kavg.py <rocprof dir> <substring> ... : calls and average microseconds of the kernels whose name contains a substring."""
import csv, glob, sys
d, subs = sys.argv[1], sys.argv[2:]
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(s in r["Name"] for s in subs):
            print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1e3:7.1f} us")
