"""Developer probe: PhraseConvPool fwd / fwd+bwd at the bench shape (B=160, T=26, E=512), HIP path vs
the stock torch modules (MIOpen Conv1d) of the same module object."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqa_amd
from vqa_amd.modules import PhraseConvPool

dev = torch.device("cuda", 0)
torch.manual_seed(0)
mod = PhraseConvPool(512).to(dev)
x = torch.randn(160, 26, 512, device=dev, requires_grad=True)
g = torch.randn(160, 26, 512, device=dev)
flop_fwd = 2.0 * 160 * 26 * 512 * 512 * 6
for impl in ("hip", "stock"):
    os.environ["VQA_PHRASE_IMPL"] = impl
    def fwd():
        with torch.no_grad():
            return mod(x)
    def fb():
        for p in mod.parameters():
            p.grad = None
        x.grad = None
        mod(x).backward(g)
    for name, fn, fl in (("fwd", fwd, flop_fwd), ("fwd+bwd", fb, 3 * flop_fwd)):
        for _ in range(40):
            fn()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 100
        print("%-6s %-8s %8.3f ms  (%5.1f TFLOP/s on the %.1f GFLOP of live taps)" % (impl, name, t, fl / t / 1e9, fl / 1e9), flush=True)
