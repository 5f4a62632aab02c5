#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs of tools/probe_hot.py) into an entry of
profiles/pmc_traffic_backward.json: HBM-side bytes per launch of every kernel of coattn_backward at one shape.
FETCH_SIZE is in KiB and doubled (gfx950 halves the bytes of wide coalesced reads, MI355X_MICROARCH.md); WRITE_SIZE (KiB)
is exact.  The doubling is calibrated for 16-byte-per-lane streams; kernels with dword loads (the accumulator-shaped
fragment loads of bwd_nat32 / bwd_dq32) are uncalibrated -- read their figures as ratios between builds, not as absolutes.

usage: tools/pmc_traffic_bwd.py <fetch dir> <write dir> B N T d L layout
env: PRODUCTS (exact | fast16; default exact) = the arithmetic tools/probe_hot.py ran in: one entry per (shape, layout, products)."""
import csv
import glob
import json
import os
import re
import sys

fd, wd = sys.argv[1:3]
B, N, T, d, L = [int(x) for x in sys.argv[3:8]]
layout = sys.argv[8]
products = os.environ.get("PRODUCTS", "exact")
MARKS = {"coattn_fwd32_kernel": "coattn_fwd32", "attend_v_lm_kernel": "attend_v", "attend_v_kernel": "attend_v", "gemm_w_kernel": "projections",
         "gemm_h2p_kernel": "projections", "gemm_h2_kernel": "projections", "bwd_pre_kernel": "bwd_pre", "bwd_dc32_kernel": "bwd_dc32", "bwd_nat32_kernel": "bwd_nat32", "bwd_dq32x_kernel": "bwd_dq",
         "bwd_dq32_kernel": "bwd_dq", "gemm_tn_kernel": "bwd_gemm", "gemm_tn_wide_kernel": "bwd_gemm", "reduce_partials4_kernel": "reduce_partials"}
vals = {}
for dd, ctr in ((fd, "FETCH_SIZE"), (wd, "WRITE_SIZE")):
    for f in glob.glob(os.path.join(dd, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr:
                continue
            k = re.sub(r"^void |\(anonymous namespace\)::|[<(].*$", "", r["Kernel_Name"])
            if k in MARKS:
                vals.setdefault((MARKS[k], ctr), []).append(float(r["Counter_Value"]))
kern = {}
for (m, c), v in vals.items():
    kern.setdefault(m, {})[c] = sum(v) / len(v)
out = {"shape": {"B": B, "N": N, "T": T, "d": d, "L": L}, "layout": layout, "products": products, "kernels": {}}
for m, cs in kern.items():
    f, w = cs.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0, cs.get("WRITE_SIZE", 0.0) * 1024.0
    out["kernels"][m] = {"hbm_bytes_per_launch": int(f + w), "fetch_bytes_corrected_x2": int(f), "write_bytes": int(w)}
out["note"] = ("FETCH_SIZE x2 (gfx950, calibrated for 16-byte-per-lane streams), WRITE_SIZE exact; separate --pmc passes of "
               "tools/probe_hot.py")
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic_backward.json")
try:
    entries = json.load(open(path))["entries"]
except (OSError, ValueError, KeyError):
    entries = []
for e in entries:
    e.setdefault("products", "fast16")               # (entries of rounds 4-5: tolerance-mode runs)
entries = [e for e in entries if not (e.get("shape") == out["shape"] and e.get("layout") == layout and e["products"] == products)] + [out]
json.dump({"entries": entries}, open(path, "w"), indent=1)
print(json.dumps(out))
