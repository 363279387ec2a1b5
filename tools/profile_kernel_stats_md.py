#!/usr/bin/env python3
"""On the GPU box: condense the newest <dir>/*/*_kernel_stats.csv of a `rocprofv3 --kernel-trace --stats` run into a markdown table (stdout)."""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "*", "*_kernel_stats.csv")), key=os.path.getmtime)[-1]
print("| kernel | calls | total ns | average ns | % | min ns | max ns |\n|---|---|---|---|---|---|---|")
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    name = r["Name"].split("(")[0][:60]
    print("| `%s` | %s | %s | %.1f | %.3f | %s | %s |" % (name, r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), float(r["Percentage"]), r["MinNs"], r["MaxNs"]))
