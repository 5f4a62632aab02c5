"""Developer probe: the projections of coattn_forward (P_v, P_q) timed with HIP events; LAYOUT=lm|cm, N, B."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, vqa_amd
dev = torch.device("cuda", 0)
N = int(os.environ.get("N", "196")); B = int(os.environ.get("B", "160")); T, d = 26, 512
torch.manual_seed(0)
co = vqa_amd.ParallelCoAttention(d).to(dev)
V, Qs = bench.synth_features(B, N, T, d, dev)
x = V.permute(0, 2, 1)
if os.environ.get("LAYOUT", "lm") == "lm":
    x = x.contiguous()
args = (co.W_v.weight, co.W_v.bias, co.W_q.weight, co.W_q.bias, co.w_v.weight, co.w_v.bias, co.w_q.weight, co.w_q.bias)
with torch.no_grad():
    for _ in range(300):
        vqa_amd.coattention(x, Qs, *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        vqa_amd.coattention(x, Qs, *args)
    e1.record()
    torch.cuda.synchronize()
print("forward call %.1f us" % (e0.elapsed_time(e1) * 10))
