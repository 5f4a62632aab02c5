"""One-GPU sanity run of the RCCL ("nccl") path: process group of size 1, bucketed async all-reduce
from the autograd hooks, two training steps of the attention model."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
import torch.distributed as dist
import vqa_amd
from vqa_amd import dist as vdist, train as T
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.manual_seed(0)
model = T.build_model("attention", 100, 10).cuda()
tr = T.Trainer(model, 1e-4, torch.device("cuda:0"))
tr.reducer = vdist.GradReducer(model, bucket_mb=4.0)
b = T.synthetic_batch(8, (64, 64), 26, 100, 11, seed=1)
im, qu, la, ln = T.sort_batch(b["image"], b["question"], b["label"], b["ques_len"])
for i in range(3):
    loss = tr.step(im.cuda(), qu.cuda(), ln, la.cuda())
torch.cuda.synchronize()
print("rccl ok: loss %.4f, buckets %d, payload %.1f MB, unused %s" % (float(loss), len(tr.reducer.buckets), tr.reducer.payload_bytes() / 1e6, tr.reducer.unused))
dist.destroy_process_group()
