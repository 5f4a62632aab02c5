"""Developer probe: per-launch-group times (the library's profile marks) of coattn_forward + coattn_backward at cfg 2, pairs
rotating over three buffer sets (bench.sequence_marks); usage: probe_marks.py [N] [layout] [exact | fast]  (default exact)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from vqa_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 196
lay = sys.argv[2] if len(sys.argv) > 2 else "lm"
fast = len(sys.argv) > 3 and sys.argv[3] == "fast"
dev = torch.device("cuda", 0)
seq = bench.sequence_marks(dev, 160, N, 26, 512, 3, lay, False, fast)
print("N=%d %s %s:" % (N, lay, "fast16" if fast else "exact"), " ".join("%s %.1f" % (k, seq["avg_us"][k]) for k in seq["order"]),
      "| sum %.1f us" % sum(seq["avg_us"].values()))
