#!/usr/bin/env python3
"""Copy the summaries produced by tools/refresh_profiles.sh (gpurun_out/refresh/) into profiles/.

usage: tools/collect_profiles.py [round-tag, default r03]
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "refresh")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"


def one(pattern):
    hits = glob.glob(os.path.join(SRC, pattern), recursive=True)
    if not hits:
        raise SystemExit("missing " + pattern)
    return max(hits, key=os.path.getmtime)      # gpurun merges into gpurun_out/ without removing older runs' files


line = [l for l in open(os.path.join(SRC, "bench.json")) if l.startswith("{")][-1]
json.loads(line)
open(os.path.join(DST, "%s_bench.json" % tag), "w").write(line)
# Everything without "fast16" / "cfg4" in its name ran the reference's arithmetic (fp32-accurate products, flags = 0).
# exact_* / fast16_*: the forward kernel ALONE, launches rotating over independent buffer sets (cold: bench.py's roofline legs);
# fwd_bwd_*: coattn_forward + coattn_backward in sequence over three rotating input sets (tools/probe_hot.py)
for leg, name in (("roofline", "roofline"), ("hot", "hot_path"), ("step", "full_step"), ("bench_stats", "bench"),
                  ("iso_49_512_lm_exact", "exact_headline"), ("iso_196_512_lm_exact", "exact_reference_grid"),
                  ("iso_196_512_cm_exact", "exact_channel_major"), ("iso_49_512_lm_fast", "fast16_headline"),
                  ("iso_196_512_lm_fast", "fast16_reference_grid"), ("iso_49_2048_lm_bf16", "cfg4_headline"),
                  ("fb_49_exact", "fwd_bwd_n49"), ("fb_196_exact", "fwd_bwd_n196"),
                  ("fb_49_fast", "fast16_fwd_bwd_n49"), ("fb_196_fast", "fast16_fwd_bwd_n196")):
    shutil.copy(one("%s/**/*kernel_stats.csv" % leg), os.path.join(DST, "%s_%s_kernel_stats.csv" % (tag, name)))
shutil.copy(os.path.join(SRC, "pmc_traffic.json"), os.path.join(DST, "pmc_traffic.json"))
shutil.copy(os.path.join(SRC, "pmc_traffic_backward.json"), os.path.join(DST, "pmc_traffic_backward.json"))
s448 = [l for l in open(os.path.join(SRC, "step448.json")) if l.startswith("{")][-1]
json.loads(s448)
open(os.path.join(DST, "%s_step448_bench.json" % tag), "w").write(s448)
shutil.copy(one("step448/**/*kernel_stats.csv"), os.path.join(DST, "%s_step448_kernel_stats.csv" % tag))
c5 = [l for l in open(os.path.join(SRC, "cfg5_bench.json")) if l.startswith("{")][-1]
json.loads(c5)
open(os.path.join(DST, "%s_cfg5_bench.json" % tag), "w").write(c5)
# config 4's bench line and hot-path kernels, the answer head's kernels and its unprofiled timing
c4 = [l for l in open(os.path.join(SRC, "cfg4_bench.json")) if l.startswith("{")][-1]
json.loads(c4)
open(os.path.join(DST, "%s_cfg4_bench.json" % tag), "w").write(c4)
shutil.copy(one("cfg4_hot/**/*kernel_stats.csv"), os.path.join(DST, "%s_cfg4_hot_path_kernel_stats.csv" % tag))
shutil.copy(one("head/**/*kernel_stats.csv"), os.path.join(DST, "%s_head_kernel_stats.csv" % tag))
shutil.copy(os.path.join(SRC, "head_unprofiled.log"), os.path.join(DST, "%s_head_timing.log" % tag))

# MFMA-busy pass: mean counter value per kernel, and MFMA busy fraction = MFMA_BUSY / (32 * SQ_BUSY)
vals = {}
def newest(pattern):                                # gpurun merges accumulate older runs' files locally: newest only
    hits = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return [max(hits, key=os.path.getmtime)] if hits else []


def counter_rows(src_dir):
    """(kernel name, counter name, mean value, dispatches) of a --pmc pass: from tools/pmc_reduce.py's JSON when the box left
    one (the raw CSVs of a whole bench leg exceed what gpurun copies back), else from the newest raw CSV."""
    js = os.path.join(SRC, src_dir, "counters_by_kernel.json")
    if os.path.isfile(js):
        for k, cs in json.load(open(js)).items():
            for c, (tot, n) in cs.items():
                yield k, c, tot / max(n, 1), n
        return
    acc = {}
    for f in newest("%s/**/*counter_collection.csv" % src_dir):
        for r in csv.DictReader(open(f)):
            e = acc.setdefault((r["Kernel_Name"], r["Counter_Name"]), [0.0, 0])
            e[0] += float(r["Counter_Value"]); e[1] += 1
    for (k, c), (tot, n) in acc.items():
        yield k, c, tot / n, n


for k, c, mean, n in counter_rows("pmc_mfma"):
    kern = ("coattn_fwd32_kernel" if "coattn_fwd32" in k else "attend_v_lm_kernel" if "attend_v" in k else None)
    if kern:
        vals.setdefault(kern, {}).setdefault(c, []).append(mean)
if vals:
    out = {}
    for kern, cs in vals.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        if m.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            m["derived_mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * m["SQ_BUSY_CYCLES"])
        if m.get("SQ_WAVE_CYCLES") and "SQ_ACTIVE_INST_VALU" in m:      # share of a wave's life with a VALU/MFMA instruction issuing
            m["derived_valu_active_per_wave"] = m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"]
        out[kern] = m
    json.dump(out, open(os.path.join(DST, "%s_pmc_mfma.json" % tag), "w"), indent=1)
# the same counters over the whole isolated hot path (bench.py --only hot): one entry per kernel of the HIP library;
# pmc_cfg4_hot: config 4's hot path (the reduced-precision instantiations)
for src_dir, out_name in (("pmc_hot", "pmc_hot_path_kernels"), ("pmc_cfg4_hot", "pmc_cfg4_hot_path_kernels"),
                          ("pmc_fb_49", "pmc_fwd_bwd_n49_kernels"), ("pmc_fb_196", "pmc_fwd_bwd_n196_kernels")):
    hot, disp = {}, {}
    for k, c, mean, n in counter_rows(src_dir):
        if "anonymous namespace" not in k or "at::native" in k or "softmax_warp" in k:
            continue                                       # stock PyTorch / MIOpen / rocBLAS kernels are not ours
        name = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        hot.setdefault(name, {})[c] = mean
        disp[name] = n
    if hot:
        out = {}
        for name, cs in sorted(hot.items()):
            m = dict(cs)
            m["dispatches"] = disp[name]
            if m.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                m["derived_mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * m["SQ_BUSY_CYCLES"]), 4)
            if m.get("SQ_INSTS_MFMA") and "SQ_INSTS_VALU" in m:
                m["derived_valu_per_mfma"] = round(m["SQ_INSTS_VALU"] / m["SQ_INSTS_MFMA"], 2)
            out[name] = m
        json.dump(out, open(os.path.join(DST, "%s_%s.json" % (tag, out_name)), "w"), indent=1)
print("profiles/ refreshed from", SRC)
print(line[:400])
