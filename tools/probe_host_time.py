"""Developer probe: host time to ENQUEUE one train step (no synchronisation inside the loop) against the device
time per step -- the margin before the step becomes host-bound (matters at 8 ranks per node)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqa_amd  # noqa: E402
from vqa_amd import train as T  # noqa: E402

dev = torch.device("cuda", 0)
torch.set_num_threads(4)
torch.manual_seed(0)
model = T.build_model("attention", 10000, 1000).to(dev)
model.image_encoder.to(memory_format=torch.channels_last)
tr = T.Trainer(model, 1e-4, dev)
b = T.synthetic_batch(160, (224, 224), 26, 10000, 1001, seed=1)
im, qu, la, ln = T.sort_batch(b["image"], b["question"], b["label"], b["ques_len"])
im = im.to(dev).contiguous(memory_format=torch.channels_last)
qu, la = qu.to(dev), la.to(dev)
for _ in range(8):
    tr.step(im, qu, ln, la, next_image=im)
torch.cuda.synchronize()
n = 20
host = []
t0 = time.perf_counter()
for _ in range(n):
    h0 = time.perf_counter()
    tr.step(im, qu, ln, la, next_image=im)
    host.append(time.perf_counter() - h0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
host.sort()
print("device-paced step %.2f ms; host enqueue time per step: median %.2f ms, min %.2f ms, max %.2f ms"
      % (dt * 1e3, host[n // 2] * 1e3, host[0] * 1e3, host[-1] * 1e3))
print("(when the median is close to the step time the loop is blocked by the device queue depth, not by host work: "
      "see the min)")
