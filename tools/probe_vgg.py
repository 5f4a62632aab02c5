"""Developer probe: forward time of the stock frozen VGG11-bn encoder (B=160, 224x224, fp32)
under memory format x BatchNorm backend (MIOpen vs ATen native kernels)."""
import sys, time, os
import torch
import torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqa_amd
from vqa_amd.modules import vgg11_bn_features


class NativeBN(nn.Module):
    def __init__(self, bn):
        super().__init__()
        self.bn = bn

    def forward(self, x):
        with torch.backends.cudnn.flags(enabled=False):
            return self.bn(x)


dev = torch.device("cuda", 0)
x0 = torch.randn(160, 3, 224, 224, device=dev)
ref = None
for cl in (True, False):
    for native in (False, True):
        torch.manual_seed(0)
        m = vgg11_bn_features().to(dev)
        for p in m.parameters():
            p.requires_grad_(False)
        if native:
            for i, layer in enumerate(m):
                if isinstance(layer, nn.BatchNorm2d):
                    m[i] = NativeBN(layer)
        x = x0
        if cl:
            m = m.to(memory_format=torch.channels_last); x = x0.contiguous(memory_format=torch.channels_last)
        for _ in range(3):
            y = m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            y = m(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 100
        if ref is None:
            ref = y.float().clone()
        print("channels_last=%s native_bn=%s  fwd %.2f ms   max|diff| vs first %.2e" % (cl, native, dt, (y - ref).abs().max().item()), flush=True)
