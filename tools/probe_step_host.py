#!/usr/bin/env python3
"""Developer probe: where the HOST time of one isolated hot-path step (co-attention + answer head + loss, forward and
backward; bench.py hot_path_leg) goes -- cProfile over pipelined steps, eager and graph-replayed.
usage: tools/probe_step_host.py [N=49] [eager|static|graph]   (eager: module by module; static: one node, calls issued eagerly)"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqa_amd
from vqa_amd.modules import MLPClassifier
from vqa_amd.graph import HotPathGraph

N = int(sys.argv[1]) if len(sys.argv) > 1 else 49
mode = sys.argv[2] if len(sys.argv) > 2 else "eager"
B, T, d, K = 160, 26, 512, 1000
dev = torch.device("cuda", 0)
torch.manual_seed(0)
co = vqa_amd.ParallelCoAttention(d).to(dev)
mlp = MLPClassifier(d, 1024, K + 1).to(dev)
x_img = torch.randn(B, N, d, device=dev).clamp_min_(0)
Qs = [torch.randn(B, T, d, device=dev).requires_grad_(True) for _ in range(3)]
label = (torch.arange(B, device=dev) * 7) % (K + 1)
params = list(co.parameters()) + list(mlp.parameters())
hp = (HotPathGraph(co, mlp, B, N, T, direct_grads=True) if mode == "graph" else
      HotPathGraph(co, mlp, B, N, T, capture=False, direct_grads=True) if mode == "static" else None)

def step():
    for p in params:
        p.grad = None
    for q in Qs:
        q.grad = None
    if hp is not None:
        _, loss = hp(x_img, Qs, label)
    else:
        _, loss = mlp.forward_loss(*co(x_img, Qs), label)
    loss.backward()

for _ in range(20):
    step()
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s N=%d: host enqueue %.3f ms/step, wall %.3f ms/step" % (mode, N, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
if os.environ.get("PROFILE", "1") == "1":
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)

# ---- time spent INSIDE the C-ABI calls (hipLaunchKernel and friends) per step
if mode == "eager":
    from vqa_amd import _lib
    lib = _lib.load()
    acc = {}
    def wrap(name):
        fn = getattr(lib, name)
        def w(*a):
            t = time.perf_counter()
            r = fn(*a)
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
            return r
        setattr(lib, name, w)
    for nm in ("coattn_forward", "coattn_backward", "coattn_head_forward", "coattn_head_backward"):
        wrap(nm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("host enqueue %.3f ms/step; inside the C calls: %s" % ((t1 - t0) / n * 1e3, {k: "%.1f us" % (v / n * 1e6) for k, v in acc.items()}))
