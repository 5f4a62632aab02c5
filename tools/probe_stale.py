#!/usr/bin/env python3
"""Developer tool: the computation of tests/test_gpu_edges.py::test_large_batch_is_sample_independent with the caching
allocator's free blocks pre-filled with NaN (or a constant): a kernel that reads workspace it never wrote shows up as NaN /
as a mismatch between a batch and the sum of its halves.   usage: tools/probe_stale.py [N] [fill: nan|7|0] [layout cm|lm]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqa_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 196
fill = sys.argv[2] if len(sys.argv) > 2 else "nan"
layout = sys.argv[3] if len(sys.argv) > 3 else "cm"
dev = "cuda"
# pollute: grab most of a few size classes, fill, free (the blocks stay in torch's cache and are handed out again)
blocks = [torch.empty(n, device=dev) for n in (1 << 28, 1 << 28, 1 << 27, 1 << 27, 1 << 26, 1 << 26, 1 << 25, 1 << 25, 1 << 24, 1 << 22, 1 << 20, 1 << 18)]
for b in blocks:
    b.fill_(float("nan") if fill == "nan" else float(fill))
del blocks
torch.manual_seed(1)
B, T, d = 640, 26, 512
m = vqa_amd.ParallelCoAttention(d).to(dev)
V = torch.randn(B, d, N, device=dev).clamp_min_(0)
lens = [T] + [1 + (5 * i) % T for i in range(B - 1)]
mask = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).float().unsqueeze(-1).to(dev)
Qs = [(torch.randn(B, T, d, device=dev) * (2.0 / d) ** 0.5 * mask) for _ in range(3)]
gv = torch.randn(3, B, d, device=dev); gq = torch.randn(3, B, d, device=dev)
def run(sl):
    for p in m.parameters():
        p.grad = None
    q = [t[sl].clone().requires_grad_(True) for t in Qs]
    x = V[sl].permute(0, 2, 1)
    x = (x.clone() if layout == "cm" else x.contiguous()).requires_grad_(True)
    v_o, q_o = m(x, q)
    loss = sum((v_o[l] * gv[l, sl]).sum() + (q_o[l] * gq[l, sl]).sum() for l in range(3))
    loss.backward()
    return {"v": torch.stack(v_o).detach(), "q": torch.stack(q_o).detach(), "dV": x.grad, "dQ": torch.stack([t.grad for t in q]),
            **{"d" + n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}}
big = run(slice(0, B)); h1 = run(slice(0, 320)); h2 = run(slice(320, 640))
for k in big:
    bad = (~torch.isfinite(big[k])).sum().item() + (~torch.isfinite(h1[k])).sum().item() + (~torch.isfinite(h2[k])).sum().item()
    if k.startswith("d") and k not in ("dV", "dQ"):
        ref = h1[k] + h2[k]
        err = ((big[k] - ref).abs().max() / ref.abs().max()).item()
    else:
        ref = torch.cat([h1[k], h2[k]], dim=1 if k in ("v", "q", "dQ") else 0)
        err = ((big[k] - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
    print("%-12s nonfinite %d  rel.err(batch vs halves) %.2e" % (k, bad, err))
