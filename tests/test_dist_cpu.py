"""CPU, 2 processes over gloo: the data-parallel gradient reducer.  Averaged gradients of two
half-batches equal the single-process full-batch gradients on the co-attention + MLP subgraph
(features fixed), the dead W_b is discovered and skipped, and training steps stay in lockstep."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _build(d, K):
    from oracle.coattn_oracle import OracleMLPClassifier, OracleParallelCoAttention

    class Sub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.co_attention = OracleParallelCoAttention(d, as_executed=False)
            self.mlp_classify = OracleMLPClassifier(d, 2 * d, K)

        def forward(self, x_img, qs):
            return self.mlp_classify(*self.co_attention(x_img, qs))
    torch.manual_seed(0)
    return Sub()


def _data(B, N, T, d, K):
    from oracle import coattn_oracle as O
    V, Qs = O.make_inputs(B, N, T, d, 55, lens=[T] * B, scale_q=0.2)
    label = torch.arange(B) % K
    return V.permute(0, 2, 1), Qs, label


def _worker(rank, world, port, q, exchange="allreduce"):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from vqa_amd import dist as vdist
    vdist.init_from_env("gloo")
    B, N, T, d, K = 8, 9, 5, 32, 7
    model = _build(d, K)
    x, Qs, label = _data(B, N, T, d, K)
    sl = slice(rank * B // world, (rank + 1) * B // world)
    red = vdist.GradReducer(model, bucket_mb=0.01, exchange=exchange)   # tiny buckets: several collectives
    opt = torch.optim.Adam(model.parameters(), 1e-3)
    crit = torch.nn.CrossEntropyLoss()
    grads_step = []
    for step in range(3):
        loss = crit(model(x[sl], [qq[sl] for qq in Qs]), label[sl])
        opt.zero_grad()
        red.prepare()
        loss.backward()
        red.finish()
        grads_step.append({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
        opt.step()
    # the peer-mapped exchange needs GPU buckets: on CPU tensors every rank alike falls back to the all-reduce (no rank
    # left in a collective, no exception in the middle of a run) and records why
    nb = len(red.buckets)
    red.reset("p2p")
    for _ in range(2):                                   # the step that rebuilds the buckets, and a hooked one
        loss = crit(model(x[sl], [qq[sl] for qq in Qs]), label[sl])
        opt.zero_grad()
        red.prepare()
        loss.backward()
        local = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        red.finish()
    assert red.exchange == "allreduce" and "GPU" in red.fallback_reason
    import torch.distributed as dist
    for n, p in model.named_parameters():                # averaged: the mean of the ranks' local gradients
        if p.grad is not None:
            tot = local[n].clone()
            dist.all_reduce(tot)
            assert torch.allclose(p.grad, tot / world, atol=1e-7, rtol=1e-5), n
    try:
        vdist.GradReducer(model, exchange="none")        # a timing leg of bench.py (reset()), never a training mode
        raise AssertionError("exchange='none' must be refused by the constructor")
    except ValueError:
        pass
    # plain numpy through the queue (tensor fd-sharing dies with the worker)
    q.put((rank, red.unused, nb, {n: g.numpy() for n, g in grads_step[0].items()},
           {n: p.detach().numpy().copy() for n, p in model.named_parameters()}))
    vdist.shutdown()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("exchange", ["allreduce", "direct"])
def test_grad_reducer_world2_equals_full_batch(exchange):
    """exchange = "direct": the one-shot all-to-all / local sum / all-gather pattern."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, exchange)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, unused0, nb0, g0, w0), (_, unused1, nb1, g1, w1) = res
    g0, g1, w0, w1 = [{n: torch.from_numpy(v) for n, v in dct.items()} for dct in (g0, g1, w0, w1)]
    assert unused0 == unused1 == ["co_attention.W_b.weight", "co_attention.W_b.bias"]
    assert nb0 == nb1 and nb0 > 1
    # single-process full batch reference
    B, N, T, d, K = 8, 9, 5, 32, 7
    model = _build(d, K)
    x, Qs, label = _data(B, N, T, d, K)
    opt = torch.optim.Adam(model.parameters(), 1e-3)
    crit = torch.nn.CrossEntropyLoss()
    first = None
    for step in range(3):
        loss = crit(model(x, Qs), label)
        opt.zero_grad(); loss.backward()
        if first is None:
            first = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        opt.step()
    for n, g in first.items():
        assert torch.allclose(g0[n], g, atol=1e-6, rtol=1e-5) and torch.equal(g0[n], g1[n]), n
    for n, p in model.named_parameters():
        assert torch.equal(w0[n], w1[n]), n                                   # ranks in lockstep
        if n.endswith("w_v.bias") or n.endswith("w_q.bias"):
            continue     # analytically zero gradient: Adam turns its rounding noise into +-lr steps
        assert torch.allclose(w0[n], p.detach(), atol=1e-5), n                # == big-batch training


def _worker3(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from vqa_amd import dist as vdist
    vdist.init_from_env("gloo")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))     # 40 + 18 = 58 elements: not % 3
    out = {}
    for exchange in ("allreduce", "direct"):
        red = vdist.GradReducer(model, bucket_mb=0.0001, exchange=exchange)
        for step in range(2):                                                   # step 0 builds, step 1 runs hooked
            for i, p in enumerate(model.parameters()):
                p.grad = None
            red.prepare()
            x = torch.full((4, 7), float(rank + 1 + step))
            model(x).square().sum().backward()
            red.finish()
        out[exchange] = [p.grad.clone().numpy() for p in model.parameters()]
        for h in red._hooks:
            h.remove()
    # one reducer switched between the patterns (bench.py's timing legs): same values again; "none" leaves them local
    red = vdist.GradReducer(model, bucket_mb=0.0001, exchange="allreduce")
    for exchange in ("allreduce", "direct", "none"):
        red.reset(exchange)
        for step in range(2):
            for p in model.parameters():
                p.grad = None
            red.prepare()
            model(torch.full((4, 7), float(rank + 1 + step))).square().sum().backward()
            if exchange == "none":
                local = [p.grad.clone().numpy() for p in model.parameters()]
            red.finish()
        out["reset_" + exchange] = [p.grad.clone().numpy() for p in model.parameters()]
    out["local"] = local
    q.put((rank, out))
    vdist.shutdown()


@pytest.mark.timeout(300)
def test_direct_exchange_world3_matches_allreduce():
    """Three ranks, bucket sizes that are not multiples of the world size (padded shards): the one-shot exchange
    gives the same averaged gradients as the ring all-reduce, identical on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker3, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = res[0][1]
    for a, b in zip(ref["allreduce"], ref["direct"]):
        assert torch.allclose(torch.from_numpy(a), torch.from_numpy(b), rtol=1e-6, atol=1e-7)
    for _, out in res[1:]:
        for k in ("allreduce", "direct"):
            for a, b in zip(ref[k], out[k]):
                assert (a == b).all()                                           # ranks hold identical gradients
    for _, out in res:
        for k in ("allreduce", "direct"):
            for a, b in zip(out[k], out["reset_" + k]):
                assert (a == b).all()                                           # a re-set reducer: the same values
        for a, b in zip(out["local"], out["reset_none"]):
            assert (a == b).all()                                               # "none": local gradients, untouched


def _worker8(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from vqa_amd import dist as vdist, train as T
    vdist.init_from_env("gloo")
    torch.manual_seed(0)
    model = T.build_model("attention", 10000, 1000)      # BASELINE config 3's model: 12.18 M live trainable parameters
    live = [(n, p) for n, p in model.named_parameters() if p.requires_grad and "W_b" not in n]

    def backward(step):
        # gradient of parameter i on rank r at step s = the constant (r + 1) * (i % 7 + 1) + s: a loss that is linear in
        # the parameters gives exactly that through autograd (hooks fire as in a real backward), W_b gets none
        loss = sum((p * float((rank + 1) * (i % 7 + 1) + step)).sum() for i, (_, p) in enumerate(live))
        loss.backward()

    out = {}
    red = vdist.GradReducer(model, exchange="direct")     # default 16 MB buckets, as the trainer builds it
    for exchange in ("direct", "p2p"):                    # (p2p on CPU: falls back to all-reduce on its 4 * world padding)
        if exchange != "direct":
            red.reset(exchange)
        for step in range(2):
            for _, p in live:
                p.grad = None
            red.prepare()
            backward(step)
            red.finish()
        mean = (world + 1) / 2.0
        ok = all(torch.allclose(p.grad, torch.full_like(p, mean * (i % 7 + 1) + 1.0), rtol=1e-6)
                 for i, (_, p) in enumerate(live))
        pad = world if exchange == "direct" else 4 * world
        out[exchange] = dict(ok=ok, buckets=len(red.buckets), payload=red.payload_bytes(),
                             padded=[b.flat.numel() for b in red.buckets], pad_ok=all(b.flat.numel() % pad == 0 for b in red.buckets),
                             unused=red.unused, final=red.exchange)
    q.put((rank, out))
    vdist.shutdown()


@pytest.mark.timeout(600)
def test_world8_direct_exchange_at_the_real_payload():
    """Eight ranks (the node the scaling run uses) over gloo, the attention model's real gradient layout -- 48.7 MB in
    16 MB buckets: the one-shot exchange's shard arithmetic (bucket sizes padded to the world size; 4 x world for the
    peer-mapped variant's 16-byte shards) gives the rank mean on every rank, through the hooked path too."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=540) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for _, out in res:
        for ex in ("direct", "p2p"):
            o = out[ex]
            assert o["ok"] and o["pad_ok"], (ex, o)
            assert o["unused"] == ["co_attention.W_b.weight", "co_attention.W_b.bias"]
            assert o["payload"] == res[0][1][ex]["payload"] and 48e6 < o["payload"] < 49.5e6, o["payload"]
            assert 3 <= o["buckets"] <= 4
        assert out["direct"]["final"] == "direct" and out["p2p"]["final"] == "allreduce"


def _worker_bert(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from vqa_amd import dist as vdist
    vdist.init_from_env("gloo")
    model, batch = _bert_model_and_batch()
    sl = slice(rank * 4 // world, (rank + 1) * 4 // world)
    red = vdist.GradReducer(model, bucket_mb=0.05)
    crit = torch.nn.CrossEntropyLoss()
    for step in range(2):
        model.zero_grad(set_to_none=True)
        red.prepare()
        crit(model(batch[0][sl], batch[1][sl], batch[2][sl]), batch[3][sl]).backward()
        red.finish()
    bucketed = sorted(n for b in red.buckets for p in b.params for n, pp in model.named_parameters() if pp is p)
    q.put((rank, red.unused, bucketed, red.payload_bytes(),
           {n: p.grad.numpy().copy() for n, p in model.named_parameters() if p.grad is not None}))
    vdist.shutdown()


def _bert_model_and_batch():
    """BASELINE config 5's trainable subgraph on the CPU: QuestionBertCoAttentionEncoder (frozen BERT token embeddings ->
    Linear(768 -> d) word level, phrase and sentence levels on top) -> co-attention -> MLP, image features fixed."""
    from oracle.coattn_oracle import OracleMLPClassifier, OracleParallelCoAttention
    from vqa_amd import train as T
    from vqa_amd.modules import QuestionBertCoAttentionEncoder
    d, K, B, N, Tq = 32, 7, 4, 9, 6

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.question_encoder = QuestionBertCoAttentionEncoder(**T.bert_question_params(hidden_dim=d, vocab_size=50, bert_dim=16))
            self.co_attention = OracleParallelCoAttention(d, as_executed=False)
            self.mlp_classify = OracleMLPClassifier(d, 2 * d, K)

        def forward(self, x_img, tok, lens):
            return self.mlp_classify(*self.co_attention(x_img, list(self.question_encoder(tok, lens))))
    torch.manual_seed(0)
    net = Net()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, N, d, generator=g).clamp_min(0)
    lens = torch.tensor([Tq, Tq, Tq, Tq])                # equal lengths: every shard packs the same way
    tok = torch.randint(2, 50, (B, Tq), generator=g)
    return net, (x, tok, lens, torch.arange(B) % K)


@pytest.mark.timeout(300)
def test_config5_bert_word_level_world2():
    """BASELINE config 5 over two gloo ranks: the frozen BERT embeddings never enter a bucket (they need no gradient),
    W_b is discovered as unused, and the averaged half-batch gradients are the full-batch gradients."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bert, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    model, batch = _bert_model_and_batch()
    torch.nn.CrossEntropyLoss()(model(*batch[:3]), batch[3]).backward()
    full = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    live = sum(p.numel() for n, p in model.named_parameters() if p.requires_grad and "W_b" not in n)
    for _, unused, bucketed, payload, grads in res:
        assert unused == ["co_attention.W_b.weight", "co_attention.W_b.bias"]
        assert not any("bert" in n for n in bucketed) and any("word_proj" in n for n in bucketed)
        assert payload == 4 * live
        for n, g in full.items():
            if n.endswith("w_v.bias") or n.endswith("w_q.bias"):
                continue                                 # analytically zero (softmax shift invariance): rounding noise
            assert torch.allclose(torch.from_numpy(grads[n]), g, atol=2e-6, rtol=1e-4), n
    for n in res[0][4]:
        assert (res[0][4][n] == res[1][4][n]).all(), n
