#!/usr/bin/env python3
"""Developer tool: median duration per (kernel, grid size) from a rocprofv3 --kernel-trace csv under the given directory
(kernel_stats.csv averages over every launch of a name; a name launched at several grids needs this).
usage: tools/ktrace.py <dir> [substring ...]"""
import collections, csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
want = sys.argv[2:]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if want and not any(w in n for w in want):
        continue
    agg[(n.replace("(anonymous namespace)::", "").replace("void ", "")[:64], r.get("Grid_Size_X", r.get("Grid_Size", "")))].append(
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%-66s grid %8s  calls %5d  median %8.1f us  total %8.1f ms" % (k[0], k[1], len(v), v[len(v) // 2], sum(v) / 1e3))
