"""Developer probe: per-launch-group times (the library's profile marks) of coattn_forward + coattn_backward at cfg 2,
tolerance mode; usage: probe_marks.py [N] [layout] [exact]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from vqa_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 196
lay = sys.argv[2] if len(sys.argv) > 2 else "lm"
exact = len(sys.argv) > 3
dev = torch.device("cuda", 0)
lib, stream, fwd, bwd = bench.coattn_c_calls(dev, 160, N, 26, 512, 3, lay, False, exact)
for _ in range(100): fwd(); bwd()
us = (C.c_float * 48)(); names = C.create_string_buffer(2048)
tot, order = {}, []
R = 60
for _ in range(R):
    lib.coattn_profile_begin(stream); fwd(); bwd()
    n = lib.coattn_profile_end(us, names, 2048, 48)
    for i, nm in enumerate(names.value.decode().split("\n")[:n]):
        if nm not in tot: tot[nm] = 0.0; order.append(nm)
        tot[nm] += us[i]
print("N=%d %s %s:" % (N, lay, "exact" if exact else "tolerance"), " ".join("%s %.1f" % (k, tot[k] / R) for k in order), "| sum %.1f us" % (sum(tot.values()) / R))
