"""Developer probe: the stock encoder's first convolution (3 -> 64 channels, 224x224, B = 160, fp32, channels_last)
with its input channels zero-padded to 4 / 8 (exactly value-preserving: the extra weights are zeros)."""
import os
import sys
import time

import torch
import torch.nn.functional as F

dev = torch.device("cuda", 0)
torch.manual_seed(0)
B, S = 160, 224
x = torch.randn(B, 3, S, S, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(64, 3, 3, 3, device=dev) * 0.1


def timed(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        y = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, y


ref = None
for cin in (3, 4, 8):
    wp = F.pad(w, (0, 0, 0, 0, 0, cin - 3)).contiguous(memory_format=torch.channels_last)
    pad = lambda: F.pad(x, (0, 0, 0, 0, 0, cin - 3)).contiguous(memory_format=torch.channels_last) if cin > 3 else x
    xp = pad()
    t_conv, y = timed(lambda: F.conv2d(xp, wp, None, 1, 1))
    t_all, _ = timed(lambda: F.conv2d(pad(), wp, None, 1, 1))
    if ref is None:
        ref = y
    print("C_in = %d: conv %.3f ms, pad + conv %.3f ms, max|diff| vs C_in=3 %.2e" % (cin, t_conv, t_all, (y - ref).abs().max().item()),
          flush=True)
