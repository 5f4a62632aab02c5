#!/usr/bin/env python3
"""Developer probe: coattn_linear_forward at the shapes of the path's projections (P_v: M = 31360, P_q / dQ: M = 12480;
N = K = 512; MS=comma-separated row counts), per width: exact (flags 0), two bf16 pieces (FLAG_SPLIT2), two fp16 pieces (FLAG_F16PAIR).  HIP-event windows as bench.py's projection leg."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
d = 512
for M in [int(x) for x in os.environ.get("MS", "31360,12480,7840").split(",")]:
    x = torch.randn(M, d, device=dev); W = torch.randn(d, d, device=dev) / d ** 0.5
    y = torch.empty(M, d, device=dev); wimg = torch.empty(lib.coattn_linear_workspace_bytes(d, d) // 4, device=dev)
    widths = (("exact", 0), ("two-piece", _lib.FLAG_SPLIT2), ("two-fp16", _lib.FLAG_F16PAIR))
    for name, fl in (widths[2:] if os.environ.get("H2ONLY") else widths[:1] if os.environ.get("EXACTONLY") else widths):
        call = lambda f: lib.coattn_linear_forward(x.data_ptr(), d, W.data_ptr(), None, y.data_ptr(), wimg.data_ptr(), M, d, d, 0.0, f | fl, st)
        _lib.check(call(0), "linear")
        for _ in range(200): call(1)
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100): call(1)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 10)
        t = sorted(ts)[1]
        print("M=%5d %-9s %6.1f us  %6.1f TFLOP/s fp32-eq" % (M, name, t, 2.0 * M * d * d / t / 1e6))
