"""Developer probe: the fused forward (coattn_attention_forward) at the cfg-2 kernel shape, a few launches (for rocprofv3)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
d = torch.device("cuda", 0)
r = bench.roofline_leg(d, N=int(os.environ.get("N", "196")), iters=int(os.environ.get("ITERS", "10")), layout=os.environ.get("LAYOUT", "lm"))
print(r["avg_launch_us"])
