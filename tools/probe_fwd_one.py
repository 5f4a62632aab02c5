"""Developer probe: the fused forward (coattn_attention_forward) at one kernel shape, a few launches (for rocprofv3).
Shape from the environment: N (196), D (512), LAYOUT (lm | cm), ITERS (10); B = 160, T = 26, L = 3.
PRECISION (exact | fast; default exact: flags = 0, what bench.py's `roofline` times), COLD (1 | 0; default 1: launches rotate
over independent buffer sets as in bench.py's roofline leg; 0: one set replayed).  With COLD=1 the warm replay windows of the
leg run in the same process: SKIP_WARM=1 (default) leaves them out so that a rocprofv3 average is the cold one alone."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
cold = os.environ.get("COLD", "1") == "1"
if cold and os.environ.get("SKIP_WARM", "1") == "1":
    bench.ROOFLINE_SKIP_WARM = True
    bench.sequence_marks = lambda *a, **k: {"avg_us": {}, "order": [], "calls": 0, "buffer_sets": 0}   # (the kernel alone)
d = int(os.environ.get("D", "512"))
r = bench.roofline_leg(dev, N=int(os.environ.get("N", "196")), d=d,
                       iters=int(os.environ.get("ITERS", "10")), layout=os.environ.get("LAYOUT", "lm"),
                       fast=os.environ.get("PRECISION", "exact") == "fast", bf16=os.environ.get("OPT", "0") == "1", cold=cold)
print(r["avg_launch_us"], r["frac"], r["products"][:40])
