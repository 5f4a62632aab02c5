"""Developer probe: the fused forward (coattn_attention_forward) at one kernel shape, a few launches (for rocprofv3).
Shape from the environment: N (196), D (512), LAYOUT (lm | cm), ITERS (10); B = 160, T = 26, L = 3."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
r = bench.roofline_leg(dev, N=int(os.environ.get("N", "196")), d=int(os.environ.get("D", "512")),
                       iters=int(os.environ.get("ITERS", "10")), layout=os.environ.get("LAYOUT", "lm"))
print(r["avg_launch_us"])
