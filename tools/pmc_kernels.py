#!/usr/bin/env python3
"""Developer tool: mean counter values per kernel from rocprofv3 --pmc passes (one directory per pass).
usage: tools/pmc_kernels.py dir1 [dir2 ...]   -> table kernel x counter (mean per dispatch)"""
import csv, glob, os, re, sys
vals = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void |\(anonymous namespace\)::|\(.*$", "", r["Kernel_Name"])
            vals.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k in sorted(vals, key=lambda k: -len(vals[k])):
    if "at::" in k or "rocclr" in k:
        continue
    m = {c: sum(v) / len(v) for c, v in vals[k].items()}
    print(k[:60])
    print("   " + "  ".join("%s=%.4g" % (c.replace("SQ_", ""), v) for c, v in sorted(m.items())))
    if m.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        print("   mfma_busy=%.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * m["SQ_BUSY_CYCLES"])), end="")
    if m.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY"):
            if c in m:
                print("  %s/wave_cycles=%.3f" % (c.replace("SQ_", ""), m[c] / m["SQ_WAVE_CYCLES"]), end="")
    if m.get("SQ_INSTS_MFMA") and "SQ_INSTS_VALU" in m:
        print("  valu_per_mfma=%.2f" % ((m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]) / m["SQ_INSTS_MFMA"]), end="")
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
        print("  l2_hit=%.3f" % (m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])), end="")
    print()
