#!/usr/bin/env python3
"""Developer tool: compile csrc/<name>.hip for gfx950 to assembly (device only) and print VGPRs / scratch / LDS per kernel.
usage: tools/regs.py coattn_fwd32 gemm_tn ...   (the per-file flags of csrc/Makefile are applied to all)"""
import os, re, subprocess, sys, tempfile
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "visual-question-answering_amd", "csrc")
tmp = tempfile.mkdtemp()
for f in sys.argv[1:]:
    out = os.path.join(tmp, f + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-mllvm",
                    "-pragma-unroll-threshold=1000000", "--offload-device-only", "-S", "-o", out, os.path.join(root, f + ".hip")],
                   check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
        name, body = m.group(1), m.group(2)
        g = lambda k: (re.search(r"\.amdhsa_%s (\d+)" % k, body) or [None, "-"])[1]
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = re.sub(r"^void \(anonymous namespace\)::", "", dn)
        print("%-70s vgpr %3s acc_off %3s scratch %s lds %s" % (dn[:70], g("next_free_vgpr"), g("accum_offset"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
