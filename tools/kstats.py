"""Print the top rows of a rocprofv3 --stats kernel summary found under a directory."""
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not fs:
    sys.exit("no kernel_stats.csv under " + sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in list(csv.DictReader(open(fs[0])))[:n]:
    print(f'{r["Name"][:80]:80s} {r["Calls"]:>7s} {int(r["TotalDurationNs"])/1e6:10.3f} ms {float(r["AverageNs"])/1e3:10.1f} us {r["Percentage"]:>6s}%')
