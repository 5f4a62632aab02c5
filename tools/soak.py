"""Developer soak test: many train steps through the CLI loop's building blocks (prefetcher with fresh host
batches, encoder run-ahead, Adam); reports throughput and allocator statistics to catch leaks / stalls."""
import sys, os, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqa_amd
from vqa_amd import train as T

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
torch.set_num_threads(max(1, min(4, T.usable_cpus())))     # as train.main does
torch.manual_seed(0)
model = T.build_model("attention", 10000, 1000).to(dev)
model.image_encoder.to(memory_format=torch.channels_last)
tr = T.Trainer(model, 1e-4, dev)
pool = []
for i in range(4):                                   # 4 distinct host batches, cycled
    b = T.synthetic_batch(160, (224, 224), 26, 10000, 1001, seed=100 + i)
    image, question, label, lens = T.sort_batch(b["image"], b["question"], b["label"], b["ques_len"])
    pool.append((image, question, lens, label))
def host():
    for s in range(steps):
        yield pool[s % 4]
batches = T.DevicePrefetcher(host(), dev, True)
t0 = time.time(); losses = []
for s, (image, question, lens, label) in enumerate(batches):
    nxt, ready = batches.peek_image()
    loss = tr.step(image, question, lens, label, next_image=nxt, next_ready=ready)
    if (s + 1) % 50 == 0:
        losses.append(round(float(loss), 4))
        st = torch.cuda.memory_stats()
        print(json.dumps({"step": s + 1, "loss": losses[-1], "pairs_per_s": round(160 * (s + 1) / (time.time() - t0), 1),
                          "allocated_GB": round(st["allocated_bytes.all.current"] / 2**30, 2),
                          "reserved_GB": round(st["reserved_bytes.all.current"] / 2**30, 2),
                          "alloc_retries": st["num_alloc_retries"]}), flush=True)
torch.cuda.synchronize()
assert all(l == l for l in losses)
print("soak ok")
