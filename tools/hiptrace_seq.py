#!/usr/bin/env python3
"""Developer tool: host durations (us) of the hipLaunchKernel calls of a rocprofv3 --hip-trace run: the last N of them
(one per line group: gap since the previous launch's end / the launch's own duration)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*hip_api_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
rows = [r for r in csv.DictReader(open(f)) if r["Function"] in ("hipLaunchKernel", "hipMemcpyAsync", "hipMemsetAsync", "hipGraphLaunch")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n - 200:-200]
prev = None
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append("%s%.0f/%.0f" % ("" if r["Function"] == "hipLaunchKernel" else r["Function"][3:7] + ":", (s - prev) / 1e3 if prev else 0, (e - s) / 1e3))
    prev = e
print(" ".join(out))
