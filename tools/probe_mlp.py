"""Developer probe: MLPClassifier + cross entropy forward+backward at BASELINE config 2 (B=160, d=512, mlp=1024,
K=1000+1), HIP path (csrc/mlp.hip) vs the stock PyTorch-ROCm modules, device time per call with HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqa_amd  # noqa: E402
from vqa_amd.modules import MLPClassifier  # noqa: E402
from vqa_amd.mlp import CrossEntropyLoss  # noqa: E402

dev = torch.device("cuda", 0)
B, d, mlp, K = 160, 512, 1024, 1001
torch.manual_seed(0)
mod = MLPClassifier(d, mlp, K).to(dev)
v = torch.randn(3, B, d, device=dev, requires_grad=True)
q = torch.randn(3, B, d, device=dev, requires_grad=True)
lab = torch.randint(0, K, (B,), device=dev)
for impl in ("hip", "stock"):
    os.environ["VQA_MLP_IMPL"] = impl
    crit = CrossEntropyLoss() if impl == "hip" else torch.nn.CrossEntropyLoss()

    def step():
        mod.zero_grad(set_to_none=True)
        v.grad = q.grad = None
        loss = crit(mod([v[l] for l in range(3)], [q[l] for l in range(3)]), lab)
        loss.backward()
        return loss

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 100
    e0.record()
    for _ in range(n):
        step()
    e1.record()
    torch.cuda.synchronize()
    print("%-5s MLP + CE fwd+bwd: %.1f us per step (loss %.4f)" % (impl, e0.elapsed_time(e1) * 1e3 / n, step().item()), flush=True)
