"""Developer probe: forward time of the stock frozen VGG11-bn encoder (B=160, 224x224, fp32,
channels_last, the rewritten op graph of modules.run_conv_bn_stack) under MIOpen's solver
selection modes: immediate mode (PyTorch default) vs. measured search
(torch.backends.cudnn.benchmark = True -> miopenFindConvolutionForwardAlgorithm).
Prints per-layer conv times for both so that the slow layers are visible."""
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqa_amd  # noqa: E402
from vqa_amd.modules import vgg11_bn_features, run_conv_bn_stack  # noqa: E402

dev = torch.device("cuda", 0)
B = int(os.environ.get("B", 160))
S = int(os.environ.get("S", 224))
x0 = torch.randn(B, 3, S, S, device=dev).contiguous(memory_format=torch.channels_last)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ref = None
for bench_mode in (False, True):
    torch.backends.cudnn.benchmark = bench_mode
    torch.manual_seed(0)
    m = vgg11_bn_features().to(dev).to(memory_format=torch.channels_last)
    for p in m.parameters():
        p.requires_grad_(False)
    t_first = time.perf_counter()
    with torch.no_grad():
        y = run_conv_bn_stack(m, x0)
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t_first
    with torch.no_grad():
        ms = timed(lambda: run_conv_bn_stack(m, x0))
        y = run_conv_bn_stack(m, x0)
    if ref is None:
        ref = y.clone()
    print("cudnn.benchmark=%s: first call %.1f s, fwd %.2f ms, max|diff| vs immediate mode %.2e (|y| max %.2f)"
          % (bench_mode, t_first, ms, (y - ref).abs().max().item(), ref.abs().max().item()), flush=True)
    # per-layer conv time
    x = x0
    with torch.no_grad():
        for layer in m:
            if isinstance(layer, nn.Conv2d):
                t = timed(lambda: layer(x), 5)
                fl = 2.0 * x.shape[0] * layer.out_channels * x.shape[2] * x.shape[3] * layer.in_channels * 9
                print("   conv %4d->%4d @%3dx%3d  %.3f ms  %.1f TFLOP/s" % (layer.in_channels, layer.out_channels,
                                                                          x.shape[2], x.shape[3], t, fl / t / 1e9), flush=True)
            x = layer(x)
