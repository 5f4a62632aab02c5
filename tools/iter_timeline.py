"""Developer tool: kernels of one pipelined fwd+bwd iteration from a rocprofv3 kernel trace of `bench.py --only hot`.
usage: python tools/iter_timeline.py <dir> <grid of the iteration's first GEMM, e.g. 188416>"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"]) for r in csv.DictReader(open(f)))
idx = [i for i, r in enumerate(rows) if "gemm_w" in r[2] and r[3] == sys.argv[2]]
span, a, b = min((rows[b][0] - rows[a][0], a, b) for a, b in zip(idx, idx[1:]))
print("iteration span %.1f us, %d kernels" % (span / 1e3, b - a))
for r in rows[a - 1:b - 1]:
    print("%-64s %8s %6.1f" % (r[2][:64], r[3], (r[1] - r[0]) / 1e3))
