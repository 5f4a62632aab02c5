"""Developer probe: the stock frozen VGG11-bn forward (what bounds bench.py's `value`) under the stock framework's own
convolution-algorithm settings -- torch.backends.cudnn.benchmark off (the reference's setting: it never touches it) and on
(MIOpen's find step picks per layer).  Prints ms per forward at batch 160, 224 x 224, channels_last, fp32, and how long the
find step took.  Nothing of this repo's kernels is involved."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_amd import train as T  # noqa: E402

dev = torch.device("cuda", 0)
if os.environ.get("BENCHMARK_FIRST") == "1":      # the framework caches its choice per shape: set before the first forward
    torch.backends.cudnn.benchmark = True
    print("benchmark on from the start", flush=True)
torch.manual_seed(0)
model = T.build_model("attention", 10000, 1000).to(dev)
enc = model.image_encoder.to(memory_format=torch.channels_last)
x = torch.randn(int(os.environ.get("B", "160")), 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)


def timed(n=10):
    with torch.no_grad():
        for _ in range(3):
            y = enc(x)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(n):
            y = enc(x)
        torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3, y


base, y0 = timed()
print("benchmark off: %.2f ms per forward" % base, flush=True)
torch.backends.cudnn.benchmark = True
t0 = time.time()
with torch.no_grad():
    enc(x)
torch.cuda.synchronize()
print("find step (first forward with benchmark on): %.1f s" % (time.time() - t0), flush=True)
on, y1 = timed()
print("benchmark on:  %.2f ms per forward" % on)
print("features: max |difference| %.3e (max |feature| %.3e)" % ((y1 - y0).abs().max().item(), y0.abs().max().item()))
