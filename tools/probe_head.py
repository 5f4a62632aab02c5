"""Developer probe: device time of the answer head (MLPClassifier + CE, forward + backward) -- the HIP head behind the
C-ABI against the stock nn.Linear modules + the HIP cross entropy -- at cfg 2's shape (B=160, d=512, mlp=1024, K=1001)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqa_amd  # noqa: E402
from vqa_amd.modules import MLPClassifier  # noqa: E402

B, d, mlp, K = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (160, 512, 1024, 1001)))
dev = torch.device("cuda", 0)
torch.manual_seed(0)
mod = MLPClassifier(d, mlp, K).to(dev)
v = torch.randn(3, B, d, device=dev, requires_grad=True)
q = torch.randn(3, B, d, device=dev, requires_grad=True)
lab = (torch.arange(B, device=dev) * 7) % K
leaves = [v, q] + list(mod.parameters())


def run(impl):
    os.environ["VQA_HEAD_IMPL"] = impl

    def step():
        for t in leaves:
            t.grad = None
        logits, loss = mod.forward_loss(v, q, lab) if impl == "hip" else (None, None)
        if impl != "hip":
            logits = mod([v[l] for l in range(3)], [q[l] for l in range(3)])
            loss = vqa_amd.cross_entropy(logits, lab)
        loss.backward()

    for _ in range(30):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(200):
        step()
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 200
    return e0.elapsed_time(e1) / 200 * 1e3, wall * 1e6


for impl in ("hip", "stock", "hip", "stock"):
    dev_us, wall_us = run(impl)
    print("%-6s head fwd+bwd: %.1f us per step between events, %.1f us wall (B=%d d=%d mlp=%d K=%d)" % (impl, dev_us, wall_us, B, d, mlp, K), flush=True)

# the C-ABI alone (pre-allocated buffers, no autograd): what the launches themselves take
import ctypes as C  # noqa: E402
from vqa_amd import _lib  # noqa: E402
lib = _lib.load()
sb, wb = C.c_size_t(), C.c_size_t()
lib.coattn_head_workspace_bytes(B, d, mlp, K, 0, C.byref(sb), C.byref(wb))
saved = torch.empty(sb.value // 4, device=dev); ws = torch.empty(wb.value // 4, device=dev)
logits = torch.empty(B, K, device=dev); loss = torch.empty((), device=dev); gl = torch.ones(1, device=dev)
vd, qd = v.detach(), q.detach()
dx = torch.empty_like(vd)
ps = [p.detach() for p in (mod.W_w.weight, mod.W_w.bias, mod.W_p.weight, mod.W_p.bias, mod.W_s.weight, mod.W_s.bias, mod.W_h.weight, mod.W_h.bias)]
gs = [torch.empty_like(p) for p in ps]
rows = lambda t: (C.c_void_p * 3)(*[t[l].data_ptr() for l in range(3)])   # noqa: E731
P = _lib.HeadParams(*[t.data_ptr() for t in ps]); G = _lib.HeadParamGrads(*[t.data_ptr() for t in gs])
rv, rq, rdx = rows(vd), rows(qd), rows(dx)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def cabi(flags):
    lib.coattn_head_forward(rv, rq, C.byref(P), lab.data_ptr(), logits.data_ptr(), loss.data_ptr(), saved.data_ptr(), B, d, mlp, K, 0, flags, st)
    lib.coattn_head_backward(rv, rq, C.byref(P), saved.data_ptr(), gl.data_ptr(), None, rdx, None, C.byref(G), 0, ws.data_ptr(), B, d, mlp, K, 0, flags, st)


# interleaved rounds in one process: per-layer launches (10) against the one-launch-per-direction form (4 launches)
for _ in range(50):
    cabi(0); cabi(1)
torch.cuda.synchronize()
for rep in range(3):
    for flags, name in ((0, "per-layer launches"), (1, "one launch per direction (grid barriers)")):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(300):
            cabi(flags)
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print("C-ABI head fwd+bwd, %s: %.1f us per step between events; host enqueue %.1f us per step" % (name, e0.elapsed_time(e1) / 300 * 1e3, (t1 - t0) / 300 * 1e6), flush=True)
