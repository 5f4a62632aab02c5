"""Developer probe: forward time of the stock frozen VGG11-bn encoder (B=160, 224x224, fp32)
under memory-format x MIOpen-find settings."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqa_amd
from vqa_amd.modules import vgg11_bn_features

dev = torch.device("cuda", 0)
x0 = torch.randn(160, 3, 224, 224, device=dev)
for cl in (True, False):
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        torch.manual_seed(0)
        m = vgg11_bn_features().to(dev)
        for p in m.parameters():
            p.requires_grad_(False)
        x = x0
        if cl:
            m = m.to(memory_format=torch.channels_last); x = x0.contiguous(memory_format=torch.channels_last)
        t0 = time.perf_counter()
        for _ in range(3):
            y = m(x)
        torch.cuda.synchronize()
        tw = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(10):
            y = m(x)
        torch.cuda.synchronize()
        print("channels_last=%s benchmark=%s  fwd %.2f ms (warm-up %.1f s)" % (cl, bench, (time.perf_counter() - t0) * 100, tw), flush=True)
