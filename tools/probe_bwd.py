import sys, time, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqa_amd
dev = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 196
B, T, d = 160, 26, 512
co = vqa_amd.ParallelCoAttention(d).to(dev)
torch.manual_seed(1)
x = torch.randn(B, d, N, device=dev).clamp_min_(0).permute(0, 2, 1)
Qs = [torch.randn(B, T, d, device=dev).requires_grad_(True) for _ in range(3)]
args = (co.W_v.weight, co.W_v.bias, co.W_q.weight, co.W_q.bias, co.w_v.weight, co.w_v.bias, co.w_q.weight, co.w_q.bias)
tf = tb = 0
for it in range(25):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    v, q = vqa_amd.coattention(x, Qs, *args)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    g = [torch.ones_like(v), torch.ones_like(q)]
    torch.cuda.synchronize(); t3 = time.perf_counter()
    torch.autograd.backward([v, q], g)
    torch.cuda.synchronize(); t5 = time.perf_counter()
    if it >= 5: tf += t2 - t0; tb += t5 - t3
print("N=%d fwd %.3f ms bwd %.3f ms" % (N, tf / 20 * 1e3, tb / 20 * 1e3))
