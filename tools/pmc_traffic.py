#!/usr/bin/env python3
"""Turn rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs of
`tools/probe_fwd_one.py`: the headline shape and layout only) into an entry of profiles/pmc_traffic.json (one per shape and layout): HBM bytes per launch
of the affinity+softmax+reduce forward (coattn_fwd32_kernel + attend_v_lm_kernel / attend_v_kernel).

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE is in KiB and reports 1/2 of the
bytes of a coalesced streaming read -> doubled; the factor is calibrated on the attend_v kernel,
whose read volume is known exactly (one pass over V + a_v); WRITE_SIZE (KiB) is exact.

usage: tools/pmc_traffic.py <dir with *counter_collection.csv> [...]  B N T d L layout
env: PRODUCTS (exact | fast16 | bf16; default exact) = the arithmetic the probed launches ran in, COLD (1 | 0; default 1) =
launches rotated over independent buffer sets (tools/probe_fwd_one.py COLD=1): one entry per (shape, layout, products).
"""
import csv
import glob
import json
import os
import sys

dirs, dims, layout = sys.argv[1:-6], [int(x) for x in sys.argv[-6:-1]], sys.argv[-1]
B, N, T, d, L = dims
vals = {}
for dd in dirs:
    for f in glob.glob(os.path.join(dd, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            kern = "fwd" if "coattn_fwd32" in k else ("attend_v" if "attend_v" in k else None)
            if kern:
                vals.setdefault((kern, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
mean = {k: sum(v) / len(v) for k, v in vals.items()}
known_attend_read = 4.0 * (B * d * N + L * B * N)              # V once + a_v
# (small grids on location-major features: the forward kernel attends the image features itself, no attend_v launch)
has_attend = ("attend_v", "FETCH_SIZE") in mean
calib = known_attend_read / (mean[("attend_v", "FETCH_SIZE")] * 1024.0) if has_attend else None
fetch = (mean[("fwd", "FETCH_SIZE")] + mean.get(("attend_v", "FETCH_SIZE"), 0.0)) * 1024.0 * 2.0
write = (mean[("fwd", "WRITE_SIZE")] + mean.get(("attend_v", "WRITE_SIZE"), 0.0)) * 1024.0
products = os.environ.get("PRODUCTS", "exact")
cold = os.environ.get("COLD", "1") == "1"
out = {"shape": {"B": B, "N": N, "T": T, "d": d, "L": L}, "layout": layout, "products": products, "cold": cold,
       "hbm_bytes_per_launch": int(fetch + write),
       "fetch_bytes_corrected_x2": int(fetch), "write_bytes": int(write),
       "fetch_calibration_factor_on_attend_v": round(calib, 3) if calib else None,
       "kernels": "coattn_fwd32_kernel" + (" + attend_v kernel" if has_attend else " (attends the image features itself)"),
       "raw_kib": {"%s.%s" % k: round(v, 1) for k, v in mean.items()},
       "note": "FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 halves coalesced-read bytes); calibration on "
               "the attend_v kernel's exactly known read volume; counters from separate --pmc passes"}
# profiles/pmc_traffic.json holds one entry per (shape, layout); an entry of the same key is replaced
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
try:
    old = json.load(open(path))
    entries = old["entries"] if "entries" in old else [old]
except (OSError, ValueError):
    entries = []
for e in entries:                                   # (entries of rounds 1-5 carry no key: they were tolerance-mode / reduced-precision runs, warm)
    e.setdefault("products", "bf16" if e["shape"]["d"] == 2048 else "fast16")
    e.setdefault("cold", False)
entries = [e for e in entries if not (e.get("shape") == out["shape"] and e.get("layout") == layout and e["products"] == products)] + [out]
with open(path, "w") as fh:
    json.dump({"entries": entries}, fh, indent=1)
print(json.dumps(out))
