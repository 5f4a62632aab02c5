#!/usr/bin/env python3
"""Runs ON THE GPU BOX behind a rocprofv3 --pmc pass: reduce the pass's *counter_collection.csv (tens to hundreds of MB for a
whole bench leg; gpurun copies back at most 64 MiB) to per-kernel sums -- <dir>/counters_by_kernel.json:
{kernel name: {counter: [sum of values, dispatches]}} -- and delete the CSVs.  tools/collect_profiles.py reads the JSON.
usage: tools/pmc_reduce.py <dir> [...]"""
import csv
import glob
import json
import os
import sys

for d in sys.argv[1:]:
    acc = {}
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                e = acc.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], [0.0, 0])
                e[0] += float(r["Counter_Value"])
                e[1] += 1
    if files:
        with open(os.path.join(d, "counters_by_kernel.json"), "w") as fh:
            json.dump(acc, fh)
        for f in files:
            os.remove(f)
    print("%s: %d csv file(s) -> %d kernels" % (d, len(files), len(acc)))
