#!/usr/bin/env python3
"""Developer tool: the co-attention forward + backward alone at ONE shape (frozen image features: no dV), for a
per-kernel profile: `rocprofv3 --kernel-trace --stats -- python3 tools/probe_hot.py 196 lm 200`.
env: D (512), OPT (1 = reduced-precision mode), VQA_PRECISION (exact, the default | fast), SETS (3: iterations rotate over this
many independent input sets, so that a kernel finds in the Infinity Cache only what the kernels right before it left there),
DENSE (1: questions without pad rows; default: BASELINE's synthetic questions, lengths 3..26 of 26, pad rows zero).
Prints the wall time per iteration of the timed half."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqa_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 196
layout = sys.argv[2] if len(sys.argv) > 2 else "lm"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
B, T, d = 160, 26, int(os.environ.get("D", "512"))
dev = torch.device("cuda", 0)
co = vqa_amd.ParallelCoAttention(d).to(dev)
co.bf16_projections = os.environ.get("OPT", "0") == "1"
torch.manual_seed(1)
SETS = int(os.environ.get("SETS", "3"))
sets = []
for _ in range(SETS):
    if layout == "lm":
        x = torch.randn(B, N, d, device=dev).clamp_min_(0)
    else:
        x = torch.randn(B, d, N, device=dev).clamp_min_(0).permute(0, 2, 1)
    # questions as bench.synth_features makes them: descending lengths, rows past a length zeroed (DENSE=1: no pad rows)
    lens = torch.tensor(sorted([T] + [3 + (7 * i) % (T - 2) for i in range(B - 1)], reverse=True), device=dev)
    mask = (torch.arange(T, device=dev)[None, :] < lens[:, None]).unsqueeze(-1).float()
    if os.environ.get("DENSE", "0") == "1":
        mask = torch.ones_like(mask)
    sets.append((x, [(torch.randn(B, T, d, device=dev) * mask).requires_grad_(True) for _ in range(3)]))
g = None
k = 0
def it():
    global g, k
    x, Qs = sets[k % SETS]
    k += 1
    for q in Qs:
        q.grad = None
    for p in co.parameters():
        p.grad = None
    vs, qs = co(x, Qs)
    outs = list(vs) + list(qs)
    if g is None:
        g = [torch.ones_like(o) for o in outs]
    torch.autograd.backward(outs, g)
for _ in range(iters // 2):
    it()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    it()
torch.cuda.synchronize()
print("N=%d %s d=%d %s: %.3f ms per forward + backward (wall), %d input sets" % (N, layout, d, "fast16" if co.fast_products else ("bf16" if co.bf16_projections else "exact"), (time.perf_counter() - t0) / iters * 1e3, SETS))
