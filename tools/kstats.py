#!/usr/bin/env python3
"""Developer tool: print a rocprofv3 kernel_stats.csv (first match under the given directory) as a short table."""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True))[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for i, r in enumerate(csv.DictReader(open(f))):
    if i >= top:
        break
    print("%-72s calls %5s avg %9.1f us  %6.2f%%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
