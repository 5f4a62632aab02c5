"""Developer probe: coattn_head_forward and coattn_head_backward timed SEPARATELY, per-layer launches (flags 0) against the
one-launch-per-direction form (flags 1), at cfg 2's shape.  usage: probe_head_split.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import _lib
from vqa_amd.modules import MLPClassifier
B, d, mlp, K = 160, 512, 1024, 1001
dev = torch.device("cuda", 0)
torch.manual_seed(0)
mod = MLPClassifier(d, mlp, K).to(dev)
lib = _lib.load()
v = torch.randn(3, B, d, device=dev); q = torch.randn(3, B, d, device=dev)
lab = (torch.arange(B, device=dev) * 7) % K
sb, wb = C.c_size_t(), C.c_size_t()
lib.coattn_head_workspace_bytes(B, d, mlp, K, 0, C.byref(sb), C.byref(wb))
saved = torch.empty(sb.value // 4, device=dev); ws = torch.empty(wb.value // 4, device=dev)
logits = torch.empty(B, K, device=dev); loss = torch.empty((), device=dev); gl = torch.ones(1, device=dev)
dx = torch.empty_like(v)
ps = [p.detach() for p in (mod.W_w.weight, mod.W_w.bias, mod.W_p.weight, mod.W_p.bias, mod.W_s.weight, mod.W_s.bias, mod.W_h.weight, mod.W_h.bias)]
gs = [torch.empty_like(p) for p in ps]
rows = lambda t: (C.c_void_p * 3)(*[t[l].data_ptr() for l in range(3)])
P = _lib.HeadParams(*[t.data_ptr() for t in ps]); G = _lib.HeadParamGrads(*[t.data_ptr() for t in gs])
rv, rq, rdx = rows(v), rows(q), rows(dx)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
fwd = lambda f: lib.coattn_head_forward(rv, rq, C.byref(P), lab.data_ptr(), logits.data_ptr(), loss.data_ptr(), saved.data_ptr(), B, d, mlp, K, 0, f, st)
bwd = lambda f: lib.coattn_head_backward(rv, rq, C.byref(P), saved.data_ptr(), gl.data_ptr(), None, rdx, None, C.byref(G), 0, ws.data_ptr(), B, d, mlp, K, 0, f, st)
for f in (0, 1):
    fwd(f); bwd(f)
torch.cuda.synchronize()
def t(fn, n=300):
    for _ in range(50): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(2):
    for f, name in ((0, "per-layer launches"), (1, "one launch")):
        print("%-20s forward (+ loss) %.1f us, backward %.1f us" % (name, t(lambda: fwd(f)), t(lambda: bwd(f))), flush=True)
