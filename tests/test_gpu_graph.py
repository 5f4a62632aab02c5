"""GPU: the hot path replayed from a captured HIP graph (graph.py): coattn_forward + answer head +
their backward captured once, replayed 50 times -- every output bit for bit what the same C-ABI calls give when they
are launched eagerly, also after the inputs and the parameters have changed in place; and the whole train step with
``Trainer(graph=True)`` against the eager step."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _modules(d, mlp, K, seed=0):
    import vqa_amd
    from vqa_amd.modules import MLPClassifier
    torch.manual_seed(seed)
    return vqa_amd.ParallelCoAttention(d).cuda(), MLPClassifier(d, mlp, K).cuda()


@pytest.mark.parametrize("N", [49, 196])
def test_captured_hot_path_is_bitwise_the_eager_one(N):
    from vqa_amd.graph import HotPathGraph
    B, T, d, mlp, K = 160, 26, 512, 1024, 1001
    co, head = _modules(d, mlp, K)
    hp = HotPathGraph(co, head, B, N, T)
    g = torch.Generator(device="cuda").manual_seed(1)

    def fill():
        hp.V.copy_(torch.randn(B, N, d, device="cuda", generator=g).clamp_min_(0))
        for q in hp.Q:
            q.copy_(torch.randn(B, T, d, device="cuda", generator=g) * 0.1)
        hp.labels.copy_(torch.randint(0, K, (B,), device="cuda", generator=g))

    def outputs():
        return [t.clone() for t in [hp.logits, hp.loss, hp.v, hp.q, hp.dx] + hp.dQ + hp.co_grads + hp.head_grads]

    fill()
    hp.run_eager()
    ref = outputs()
    assert all(torch.isfinite(t).all() for t in ref)
    for _ in range(50):
        for t in [hp.logits, hp.loss] + hp.dQ + hp.co_grads + hp.head_grads:
            t.fill_(float("nan"))
        hp.replay()
        assert all(torch.equal(a, b) for a, b in zip(ref, outputs()))
    # new inputs, parameters updated in place (what Adam does): the graph reads them where they lie
    fill()
    with torch.no_grad():
        for p in hp.co_params + hp.head_params:
            p.mul_(1.01)
    hp.run_eager()
    ref2 = outputs()
    assert not torch.equal(ref2[0], ref[0])
    hp.replay()
    assert all(torch.equal(a, b) for a, b in zip(ref2, outputs()))


def test_graphed_function_matches_the_autograd_path():
    """HotPathGraph.__call__ (autograd-aware) against co_attention -> mlp_classify.forward_loss through autograd:
    same loss and gradients (upstream gradient != 1: the captured gradients are scaled)."""
    from vqa_amd.graph import HotPathGraph
    B, N, T, d, mlp, K = 12, 49, 26, 256, 128, 19
    co, head = _modules(d, mlp, K, seed=3)
    x = torch.randn(B, N, d, device="cuda").clamp_min_(0)
    Qs = [(torch.randn(B, T, d, device="cuda") * 0.2).requires_grad_(True) for _ in range(3)]
    lab = torch.arange(B, device="cuda") % K
    params = list(co.parameters()) + list(head.parameters())

    def grads():
        return [q.grad.clone() for q in Qs] + [p.grad.clone() for p in params if p.grad is not None]

    _, loss = head.forward_loss(*co(x, Qs), lab)
    (loss * 3.0).backward()
    ref, lref = grads(), loss.detach().clone()
    for t in Qs + params:
        t.grad = None
    hp = HotPathGraph(co, head, B, N, T)
    logits, loss2 = hp(x, Qs, lab)
    (loss2 * 3.0).backward()
    assert torch.equal(loss2, lref)
    for a, b in zip(ref, grads()):
        # (3 * g: one rounding apart; the biases of w_v / w_q have analytically zero gradients: rounding noise ~1e-8)
        assert (a - b).abs().max() <= 1e-6 * a.abs().max().item() + 1e-7
    # inputs at other addresses: a second graph pair is captured; a layout the kernels do not run on (channel-major rows
    # of 49 floats) is re-laid once, as on the eager path, and read from that buffer: one more address set
    x2 = x.clone()
    _, l3 = hp(x2, [q.detach().clone() for q in Qs], lab)
    _, l4 = hp(x.permute(0, 2, 1).contiguous().permute(0, 2, 1), [q.detach() for q in Qs], lab)
    assert torch.equal(l3, lref) and torch.equal(l4, lref) and len(hp._pairs) == 4
    # float64 question features cannot be read in place: they go through the static inputs (no new pair)
    _, l5 = hp(x, [q.detach().double() for q in Qs], lab)
    assert torch.equal(l5, lref) and len(hp._pairs) == 4
    assert co.W_b.weight.grad is None


def test_trainer_graph_mode_matches_eager_training():
    """Three Adam steps of the attention model with Trainer(graph=True) against the eager trainer: same losses."""
    from vqa_amd import train as T
    dev = torch.device("cuda:0")
    losses = {}
    for mode in (False, True):
        torch.manual_seed(0)
        model = T.build_model("attention", 100, 10).to(dev)
        tr = T.Trainer(model, 1e-4, dev, graph=mode)
        b = T.synthetic_batch(8, (64, 64), 26, 100, 11, seed=1)
        im, qu, la, ln = T.sort_batch(b["image"], b["question"], b["label"], b["ques_len"])
        im, qu, la = im.to(dev), qu.to(dev), la.to(dev)
        losses[mode] = [float(tr.step(im, qu, ln, la, next_image=im)) for _ in range(3)]
    assert model.hot_path_graph and len(model._graphs) == 1
    for a, b in zip(losses[False], losses[True]):
        assert abs(a - b) <= 1e-5 * abs(a)


def test_channel_major_features_are_captured_in_place(monkeypatch):
    """The reference's layout -- x_img = the permuted view of a [B,d,N] buffer (model.py:215-217) -- is read where it
    lies through the C-ABI's strides (no copy into the static input), bit for bit the eager module path.  (With
    VQA_CM_FEATURES=inplace: by default the host converts frozen channel-major features, the next test.)"""
    from vqa_amd.graph import HotPathGraph
    import sys
    ca = sys.modules["vqa_amd.coattention"]                   # (the package attribute of that name is the function)
    monkeypatch.setattr(ca, "CM_FEATURES", "inplace")
    B, N, T, d, mlp, K = 16, 196, 26, 512, 256, 37
    co, head = _modules(d, mlp, K, seed=5)
    buf = torch.randn(B, d, N, device="cuda").clamp_min_(0)
    x = buf.permute(0, 2, 1)                                 # [B,N,d], strides (d N, 1, N)
    Qs = [(torch.randn(B, T, d, device="cuda") * 0.2).requires_grad_(True) for _ in range(3)]
    lab = torch.arange(B, device="cuda") % K
    params = list(co.parameters()) + list(head.parameters())
    _, loss = head.forward_loss(*co(x, Qs), lab)
    loss.backward()
    ref = [q.grad.clone() for q in Qs] + [p.grad.clone() for p in params if p.grad is not None]
    for t in Qs + params:
        t.grad = None
    hp = HotPathGraph(co, head, B, N, T)
    hp.V.fill_(float("nan"))                                 # the static input must not be what the graph reads
    _, loss2 = hp(x, Qs, lab)
    loss2.backward()
    assert any(k[0] == buf.data_ptr() for k in hp._pairs), "the channel-major view was not captured in place"
    assert torch.equal(loss2, loss.detach())
    for a, b in zip(ref, [q.grad for q in Qs] + [p.grad for p in params if p.grad is not None]):
        assert torch.equal(a, b)


def test_frozen_channel_major_features_are_converted_by_default():
    """Frozen channel-major features go through the library's one-pass conversion into the node's static input and run on
    the location-major kernels (faster by more than the pass costs): bit for bit the run on location-major features."""
    from vqa_amd.graph import HotPathGraph
    B, N, T, d, mlp, K = 16, 196, 26, 512, 256, 37
    co, head = _modules(d, mlp, K, seed=6)
    buf = torch.randn(B, d, N, device="cuda").clamp_min_(0)
    lab = torch.arange(B, device="cuda") % K
    params = list(co.parameters()) + list(head.parameters())
    out = []
    for x in (buf.permute(0, 2, 1), buf.permute(0, 2, 1).contiguous()):
        for p in params:
            p.grad = None
        Qs = [(torch.randn(B, T, d, device="cuda", generator=torch.Generator("cuda").manual_seed(3 + l)) * 0.2).requires_grad_(True)
              for l in range(3)]
        hp = HotPathGraph(co, head, B, N, T, capture=False)
        _, loss = hp(x, Qs, lab)
        loss.backward()
        out.append([loss.detach().clone()] + [q.grad.clone() for q in Qs] + [p.grad.clone() for p in params if p.grad is not None])
        if x.stride(2) != 1:
            assert torch.equal(hp.V, x.contiguous())
    for a, b in zip(*out):
        assert torch.equal(a, b)
    _, qs = co(buf.permute(0, 2, 1), [torch.zeros(B, T, d, device="cuda") for _ in range(3)])   # the module path converts too
    assert all(torch.isfinite(t).all() for t in qs)


def test_graph_path_label_checks():
    """int32 labels are converted through the static input (the kernels read int64 at the captured address), float
    labels are refused, and a label outside [0, K) is reported by check_labels() as on the eager path."""
    from vqa_amd import head as H
    from vqa_amd.graph import HotPathGraph
    B, N, T, d, mlp, K = 8, 49, 26, 256, 128, 11
    co, head = _modules(d, mlp, K, seed=7)
    x = torch.randn(B, N, d, device="cuda").clamp_min_(0)
    Qs = [torch.randn(B, T, d, device="cuda") * 0.2 for _ in range(3)]
    lab = torch.arange(B, device="cuda") % K
    hp = HotPathGraph(co, head, B, N, T)
    _, l64 = hp(x, Qs, lab)
    _, l32 = hp(x, Qs, lab.to(torch.int32))
    assert torch.equal(l64, l32) and torch.isfinite(l64)
    with pytest.raises(RuntimeError, match="integer class indices"):
        hp(x, Qs, lab.float())
    H.check_labels()                                         # in range: nothing to report
    bad = lab.clone()
    bad[3] = K + 5
    _, lbad = hp(x, Qs, bad)
    assert torch.isnan(lbad)
    with pytest.raises(IndexError):
        H.check_labels()


def test_capture_while_another_thread_allocates():
    """train.DevicePrefetcher's worker pins and copies batches while the first step captures its graphs: the capture
    runs in thread_local error mode, so those allocator / copy calls do not invalidate it."""
    import threading
    from vqa_amd.graph import HotPathGraph
    B, N, T, d, mlp, K = 8, 49, 26, 256, 128, 11
    co, head = _modules(d, mlp, K, seed=9)
    stop = threading.Event()

    def worker():
        s = torch.cuda.Stream()
        while not stop.is_set():
            h = torch.empty(1 << 16).pin_memory()
            with torch.cuda.stream(s):
                h.to("cuda", non_blocking=True)
            s.synchronize()

    th = threading.Thread(target=worker)
    th.start()
    try:
        x = torch.randn(B, N, d, device="cuda").clamp_min_(0)
        Qs = [torch.randn(B, T, d, device="cuda") * 0.2 for _ in range(3)]
        lab = torch.arange(B, device="cuda") % K
        for _ in range(4):                                   # several captures (new address sets) under the noise
            hp = HotPathGraph(co, head, B, N, T)
            _, loss = hp(x.clone(), [q.clone() for q in Qs], lab.clone())
            assert torch.isfinite(loss)
    finally:
        stop.set()
        th.join()


def test_eager_static_mode_and_direct_grads():
    """HotPathGraph(capture=False): the same node issued eagerly -- bit for bit the captured replay and the module path;
    direct_grads=True: the static gradient buffers become param.grad (no AccumulateGrad clones), accumulate onto an
    existing .grad, and survive zero_grad(set_to_none=False)."""
    from vqa_amd.graph import HotPathGraph
    B, N, T, d, mlp, K = 12, 49, 26, 256, 128, 19
    co, head = _modules(d, mlp, K, seed=11)
    x = torch.randn(B, N, d, device="cuda").clamp_min_(0)
    Qs = [(torch.randn(B, T, d, device="cuda") * 0.2).requires_grad_(True) for _ in range(3)]
    lab = torch.arange(B, device="cuda") % K
    params = [p for p in list(co.parameters()) + list(head.parameters())]
    live = [p for p in params if p is not co.W_b.weight and p is not co.W_b.bias]

    def clear():
        for t in Qs + params:
            t.grad = None

    _, loss = head.forward_loss(*co(x, Qs), lab)
    loss.backward()
    ref = [t.grad.clone() for t in Qs + live]
    clear()
    he = HotPathGraph(co, head, B, N, T, capture=False, direct_grads=True)
    _, l1 = he(x, Qs, lab)
    l1.backward()
    assert torch.equal(l1, loss.detach())
    for a, t in zip(ref, Qs + live):
        assert torch.equal(a, t.grad)
    assert co.W_v.weight.grad is he.co_grads[0] and co.W_b.weight.grad is None      # the static buffer itself
    # a second backward without clearing: the buffers hold the new gradient in place (as after zero_grad(set_to_none=False))
    for p in live:
        p.grad.zero_()
    for q in Qs:
        q.grad = None
    _, l2 = he(x, Qs, lab)
    l2.backward()
    for a, t in zip(ref, Qs + live):
        assert torch.equal(a, t.grad)
    # gradients held elsewhere are accumulated onto
    clear()
    for p in live:
        p.grad = torch.ones_like(p)
    _, l3 = he(x, Qs, lab)
    l3.backward()
    for a, t in zip(ref[3:], live):
        assert torch.allclose(t.grad, a + 1.0, rtol=0, atol=1e-6)
    # the captured replay gives the same values
    clear()
    hg = HotPathGraph(co, head, B, N, T, direct_grads=True)
    _, l4 = hg(x, Qs, lab)
    l4.backward()
    assert torch.equal(l4, loss.detach())
    for a, t in zip(ref, Qs + live):
        assert torch.equal(a, t.grad)


def test_trainer_default_path_matches_the_module_path(monkeypatch):
    """Three Adam steps: Trainer's default (static hot path, direct gradients) against VQA_HOT_PATH=modules."""
    from vqa_amd import train as T
    dev = torch.device("cuda:0")
    losses = {}
    for mode in ("modules", "static"):
        monkeypatch.setenv("VQA_HOT_PATH", mode)
        torch.manual_seed(0)
        model = T.build_model("attention", 100, 10).to(dev)
        tr = T.Trainer(model, 1e-4, dev)
        assert model.hot_path_static == (mode == "static")
        b = T.synthetic_batch(8, (64, 64), 26, 100, 11, seed=1)
        im, qu, la, ln = T.sort_batch(b["image"], b["question"], b["label"], b["ques_len"])
        im, qu, la = im.to(dev), qu.to(dev), la.to(dev)
        losses[mode] = [float(tr.step(im, qu, ln, la, next_image=im).detach()) for _ in range(3)]
    for a, b in zip(losses["modules"], losses["static"]):
        assert abs(a - b) <= 1e-5 * abs(a)


def test_hot_path_node_is_rebuilt_when_the_module_moves():
    """The node reads the parameters where they lie; moving the module (new storages) must not leave it reading the old ones."""
    from vqa_amd import train as T
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = T.build_model("attention", 100, 10).to(dev)
    tr = T.Trainer(model, 1e-4, dev)
    b = T.synthetic_batch(4, (64, 64), 26, 100, 11, seed=1)
    im, qu, la, ln = T.sort_batch(b["image"], b["question"], b["label"], b["ques_len"])
    im, qu, la = im.to(dev), qu.to(dev), la.to(dev)
    tr.step(im, qu, ln, la)
    first = next(iter(model._graphs.values()))
    keep = [p.data for p in model.parameters()]              # (held: the allocator must not hand the same blocks back)
    model.float().cpu().to(dev)                              # round trip: every parameter gets new storage
    assert keep[0].data_ptr() != next(model.parameters()).data_ptr()
    tr.optimizer = torch.optim.Adam(model.parameters(), 1e-4)
    loss = tr.step(im, qu, ln, la)
    assert torch.isfinite(loss)
    second = next(iter(model._graphs.values()))
    assert second is not first and second.co_params[0] is model.co_attention.W_v.weight
