"""Average duration of the kernels whose name contains a pattern, from a `rocprofv3 --kernel-trace --stats -d <dir>` run:
   python tools/kstats.py <dir> <pattern> [label]"""
import csv
import glob
import sys

d, pat = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else ""
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if pat in row["Name"]:
            print(f"{label:24s} {row['Name'][:70]:70s} calls {row['Calls']:>5s}  avg {float(row['AverageNs']) / 1e3:8.1f} us")
