"""Developer probe: coattn_linear_forward in the two-FP16-piece mode (gemm_h2.hip) against the float64 product: error map by
row block / column block, and run-to-run repeatability.  usage: probe_h2.py [M N K]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, vqa_amd
from vqa_amd import _lib
lib = _lib.load()
M, N, K = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (1024, 512, 512)))
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn(M, K, generator=g).to(dev); W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev); b = torch.randn(N, generator=g).to(dev)
wimg = torch.empty(lib.coattn_linear_workspace_bytes(N, K) // 4, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ref = x.double() @ W.double().t() + b.double()
outs = []
for rep in range(3):
    y = torch.full((M, N), float("nan"), device=dev)
    rc = lib.coattn_linear_forward(x.data_ptr(), K, W.data_ptr(), b.data_ptr(), y.data_ptr(), wimg.data_ptr(), M, N, K, 0.0, _lib.FLAG_F16PAIR, st)
    torch.cuda.synchronize()
    assert rc == 0, lib.coattn_last_error()
    outs.append(y)
    e = (y.double() - ref).abs()
    print("rep", rep, "max err %.3e" % e.max().item(), "nan", int(torch.isnan(y).sum()))
    if rep == 0:
        eb = e.view(-1, 1, N)[: (M // 32) * 32].view(M // 32, 32, N // 32, 32).amax((1, 3))
        bad = (eb > 1e-4).nonzero()
        print("bad 32x32 blocks:", len(bad), "of", eb.numel(), bad[:24].tolist())
print("repeatable:", all(torch.equal(outs[0], o) for o in outs[1:]))
