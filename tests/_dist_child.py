"""Child process of tests/test_gpu_dist.py: one rank of an RCCL ("nccl") process group on cuda:LOCAL_RANK.

    python tests/_dist_child.py --mode {net,sub} --world W --rank R --port P --exchange {none,allreduce,direct,p2p} --out F
                                [--backend gloo --same-gpu]

mode net: two `Trainer.step`s of the full attention model (frozen VGG + question encoder + HIP co-attention + MLP)
          with a `GradReducer` attached (exchange none: no process group, no reducer -- the reference run);
mode sub: the co-attention + MLP subgraph on fixed features (HIP path), rank r takes shard r of the batch; first-step
          averaged gradients and the parameters after 3 Adam steps.
Writes an .npz of float32 arrays."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sub_model(d, K):
    import torch
    import vqa_amd
    from vqa_amd.modules import MLPClassifier

    class Sub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.co_attention = vqa_amd.ParallelCoAttention(d)
            self.mlp_classify = MLPClassifier(d, 2 * d, K)

        def forward(self, x_img, qs):
            return self.mlp_classify(*self.co_attention(x_img, qs))
    torch.manual_seed(0)
    return Sub()


def sub_data(B, N, T, d, K):
    import torch
    from oracle import coattn_oracle as O
    V, Qs = O.make_inputs(B, N, T, d, 55, lens=[T] * B, scale_q=0.2)
    return V.permute(0, 2, 1), Qs, torch.arange(B) % K


SUB_SHAPE = dict(B=8, N=49, T=26, d=256, K=7)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", required=True)
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--port", type=int, default=29541)
    ap.add_argument("--exchange", default="allreduce")
    ap.add_argument("--out", required=True)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--same-gpu", action="store_true", help="every rank on cuda:0 (gloo; the p2p exchange over IPC on one GPU)")
    a = ap.parse_args()
    os.environ.update(RANK=str(a.rank), WORLD_SIZE=str(a.world), LOCAL_RANK=str(a.rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(a.port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import numpy as np
    import torch
    import torch.distributed as dist
    from vqa_amd import dist as vdist, train as T
    dev = torch.device("cuda", 0 if a.same_gpu else a.rank)
    torch.cuda.set_device(dev)
    if os.environ.get("VQA_TEST_P2P_FAIL_RANK") == str(a.rank):
        # fault injection lives HERE, not in the product: on this rank exporting a bucket's IPC handle fails
        import torch.multiprocessing.reductions as _red

        def _fail(_t):
            raise RuntimeError("export failure injected by the test (VQA_TEST_P2P_FAIL_RANK)")
        _red.reduce_tensor = _fail
    if a.exchange != "none" and a.backend == "nccl":
        dist.init_process_group("nccl", rank=a.rank, world_size=a.world, device_id=dev)
    elif a.exchange != "none":
        dist.init_process_group(a.backend, rank=a.rank, world_size=a.world)
    out = {}
    if a.mode == "net":
        torch.manual_seed(0)
        model = T.build_model("attention", 100, 10).to(dev)
        tr = T.Trainer(model, 1e-4, dev)
        if a.exchange != "none":
            tr.reducer = vdist.GradReducer(model, bucket_mb=4.0, exchange=a.exchange)
        b = T.synthetic_batch(8, (64, 64), 26, 100, 11, seed=1)
        im, qu, la, ln = T.sort_batch(b["image"], b["question"], b["label"], b["ques_len"])
        im, qu, la = im.to(dev), qu.to(dev), la.to(dev)
        losses = [float(tr.step(im, qu, ln, la, next_image=im)) for _ in range(2)]
        torch.cuda.synchronize()
        out["losses"] = np.asarray(losses, dtype=np.float64)
        for n, p in model.named_parameters():
            if p.requires_grad:
                out["p." + n] = p.detach().float().cpu().numpy()
        if tr.reducer is not None:
            out["n_buckets"] = np.asarray([len(tr.reducer.buckets)])
            out["payload"] = np.asarray([tr.reducer.payload_bytes()])
            assert tr.reducer.unused == ["co_attention.W_b.weight", "co_attention.W_b.bias"], tr.reducer.unused
    else:
        s = SUB_SHAPE
        model = sub_model(s["d"], s["K"]).to(dev)
        x, Qs, label = sub_data(s["B"], s["N"], s["T"], s["d"], s["K"])
        sl = slice(a.rank * s["B"] // a.world, (a.rank + 1) * s["B"] // a.world)
        x, Qs, label = x[sl].to(dev), [q[sl].to(dev) for q in Qs], label[sl].to(dev)
        red = vdist.GradReducer(model, bucket_mb=0.25, exchange=a.exchange) if a.exchange != "none" else None
        opt = torch.optim.Adam(model.parameters(), 1e-3)
        crit = torch.nn.CrossEntropyLoss()
        for step in range(3):
            loss = crit(model(x, Qs), label)
            opt.zero_grad()
            if red is not None:
                red.prepare()
            loss.backward()
            if red is not None:
                red.finish()
            if step == 0:
                for n, p in model.named_parameters():
                    if p.grad is not None:
                        out["g." + n] = p.grad.detach().cpu().numpy()
            opt.step()
        torch.cuda.synchronize()
        for n, p in model.named_parameters():
            out["p." + n] = p.detach().cpu().numpy()
        if red is not None:
            out["n_buckets"] = np.asarray([len(red.buckets)])
            out["exchange_used"] = np.asarray([red.exchange])
            out["fallback"] = np.asarray([red.fallback_reason or ""])
            red.reset("none")                                          # (p2p: unmaps the peers' buckets, in step)
    np.savez(a.out, **out)
    if a.exchange != "none":
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
