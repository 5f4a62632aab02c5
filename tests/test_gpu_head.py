"""GPU: the answer head behind the C-ABI (coattn_head_forward / coattn_head_backward, csrc/head.hip)
against the oracle's float64 restatement of reference model.py:400-434 + nn.CrossEntropyLoss (main.py:94 / :214):
logits, loss and all input / parameter gradients.  Tolerance: north_star's 1e-4 (fp32); observed ~1e-6."""
import ctypes as C

import pytest
import torch

from oracle import coattn_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4
NAMES = ("W_w.weight", "W_w.bias", "W_p.weight", "W_p.bias", "W_s.weight", "W_s.bias", "W_h.weight", "W_h.bias")


def _rel(a, b):
    return ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _case(B, d, mlp, K, seed=0, vscale=1.0):
    torch.manual_seed(1000 + B + d + seed)
    ref = O.OracleMLPClassifier(d, mlp, K).double()
    v = torch.from_numpy(O.hash_normal((3, B, d), 11 + seed, vscale))
    q = torch.from_numpy(O.hash_normal((3, B, d), 12 + seed, 0.5))
    labels = torch.from_numpy((O.hash_uniform(B, 13 + seed) * K).astype("int64")).clamp_(0, K - 1)
    return ref, v, q, labels


def _oracle(ref, v, q, labels, g_loss=1.0, g_logits=None):
    vr, qr = v.clone().requires_grad_(True), q.clone().requires_grad_(True)
    for p in ref.parameters():
        p.grad = None
    z = ref([vr[l] for l in range(3)], [qr[l] for l in range(3)])
    loss = torch.nn.functional.cross_entropy(z, labels)
    tot = g_loss * loss
    if g_logits is not None:
        tot = tot + (z * g_logits).sum()
    tot.backward()
    return z.detach(), loss.detach(), vr.grad, qr.grad, {k: p.grad.clone() for k, p in ref.named_parameters()}


def _call(v, q, P, labels, g_loss=None, g_logits=None, separate_dq=False, accumulate=0, grads_init=None, want_dx=True,
          flags=0):
    """Straight through the C-ABI.  v, q [3,B,d]; P: dict of reference-named parameters (any device)."""
    from vqa_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    v, q = v.float().to(dev).contiguous(), q.float().to(dev).contiguous()
    ps = [P[k].float().to(dev).contiguous() for k in NAMES]
    _, B, d = v.shape
    mlp, K = ps[4].shape[0], ps[6].shape[0]
    sb, wb = C.c_size_t(), C.c_size_t()
    _lib.check(lib.coattn_head_workspace_bytes(B, d, mlp, K, _lib.F32, C.byref(sb), C.byref(wb)), "ws")
    saved = torch.full((sb.value // 4,), float("nan"), device=dev)
    logits = torch.full((B, K), float("nan"), device=dev)
    loss = torch.full((), float("nan"), device=dev) if labels is not None else None
    lab = labels.to(dev) if labels is not None else None
    rows = lambda t: (C.c_void_p * 3)(*[t[l].data_ptr() for l in range(3)])   # noqa: E731
    p = _lib.HeadParams(*[t.data_ptr() for t in ps])
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.coattn_head_forward(rows(v), rows(q), C.byref(p), lab.data_ptr() if lab is not None else None,
                                       logits.data_ptr(), loss.data_ptr() if loss is not None else None, saved.data_ptr(),
                                       B, d, mlp, K, _lib.F32, flags, stream), "coattn_head_forward")
    torch.cuda.synchronize()
    out = {"logits": logits, "loss": loss}
    if g_loss is None and g_logits is None:
        return out
    ws = torch.full((wb.value // 4,), float("nan"), device=dev)
    dv = torch.full_like(v, float("nan")) if want_dx else None
    dq = torch.full_like(q, float("nan")) if (separate_dq and want_dx) else None
    grads = [g.float().to(dev).clone() for g in grads_init] if grads_init else [torch.full_like(t, float("nan")) for t in ps]
    pg = _lib.HeadParamGrads(*[t.data_ptr() for t in grads])
    gl = torch.tensor([g_loss], device=dev, dtype=torch.float32) if g_loss is not None else None
    gx = g_logits.float().to(dev).contiguous() if g_logits is not None else None
    _lib.check(lib.coattn_head_backward(rows(v), rows(q), C.byref(p), saved.data_ptr(), gl.data_ptr() if gl is not None else None,
                                        gx.data_ptr() if gx is not None else None, rows(dv) if want_dx else None,
                                        rows(dq) if dq is not None else None, C.byref(pg), accumulate, ws.data_ptr(),
                                        B, d, mlp, K, _lib.F32, flags, stream), "coattn_head_backward")
    torch.cuda.synchronize()
    out.update({"dv": dv, "dq": dq})
    out.update({"d" + k: g for k, g in zip(NAMES, grads)})
    return out


SHAPES = [(160, 512, 1024, 1001), (5, 64, 96, 7), (3, 20, 12, 5), (33, 64, 64, 12), (7, 2048, 1024, 3001), (1, 32, 32, 2),
          (70, 96, 160, 33)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "B%d_d%d_mlp%d_K%d" % s)
def test_head_vs_oracle(shape):
    """logits, loss, d(q+v) and the eight parameter gradients against the float64 oracle; cfg 2's shape, cfg 4's
    (d = 2048, K = 3001), shapes with ragged tiles and unaligned rows (the per-element staging path)."""
    B, d, mlp, K = shape
    ref, v, q, labels = _case(B, d, mlp, K)
    z, loss, gv, gq, gp = _oracle(ref, v, q, labels, g_loss=1.7)
    r = _call(v, q, ref.state_dict(), labels, g_loss=1.7, separate_dq=True)
    assert (r["logits"].double().cpu() - z).abs().max() < TOL
    assert abs(r["loss"].item() - loss.item()) < TOL
    assert _rel(r["dv"], gv) < TOL and _rel(r["dq"], gq) < TOL
    assert torch.equal(r["dv"], r["dq"])                     # d(q_l + v_l) goes to both
    for k in NAMES:
        assert _rel(r["d" + k], gp[k]) < TOL, k


@pytest.mark.parametrize("shape", [(160, 512, 1024, 1001), (3, 20, 12, 5), (70, 96, 160, 33), (300, 64, 64, 12)],
                         ids=lambda s: "B%d_d%d_mlp%d_K%d" % s)
def test_one_launch_form_gives_the_same_bits(shape):
    """COATTN_HEAD_PERSISTENT: the layers as phases of one launch per direction behind grid-wide barriers -- the same
    tiles, so every output is bit for bit that of the per-layer launches (also when a phase has more tiles than the
    grid has workgroups, B = 300), run after run."""
    B, d, mlp, K = shape
    ref, v, q, labels = _case(B, d, mlp, K, seed=2)
    P = ref.state_dict()
    a = _call(v, q, P, labels, g_loss=1.3)
    for _ in range(4):
        b = _call(v, q, P, labels, g_loss=1.3, flags=1)
        for k in a:
            if a[k] is not None:
                assert torch.equal(a[k], b[k]), k


def test_one_launch_form_refuses_the_reduced_precision_flag():
    """The one-launch kernels exist for the exact tiles only: together with COATTN_FLAG_BF16_PROJ the call is an error, not a
    silently exact result."""
    from vqa_amd import _lib
    ref, v, q, labels = _case(8, 64, 64, 12, seed=3)
    with pytest.raises(RuntimeError, match="COATTN_HEAD_PERSISTENT has no reduced-precision form"):
        _call(v, q, ref.state_dict(), labels, g_loss=1.0, flags=1 | _lib.FLAG_BF16_PROJ)


@pytest.mark.parametrize("shape", [(160, 2048, 1024, 3001), (160, 512, 1024, 1001), (3, 20, 12, 5), (70, 96, 160, 33)],
                         ids=lambda s: "B%d_d%d_mlp%d_K%d" % s)
def test_head_reduced_precision_mode(shape):
    """flags bit 2 (COATTN_FLAG_BF16_PROJ): the operands of the four products and of their gradients rounded to bf16, one
    bf16 MFMA where the exact head issues eight f32 ones -- within bf16 tolerance of the float64 oracle (stated: 2e-2 of
    max|.| on every output, 1e-2 relative L2), not identical to the exact mode, repeatable bit for bit; config 4's shape,
    cfg 2's, and shapes on the per-element staging path with ragged tiles."""
    from vqa_amd import _lib
    B, d, mlp, K = shape
    ref, v, q, labels = _case(B, d, mlp, K, seed=4)
    z, loss, gv, gq, gp = _oracle(ref, v, q, labels, g_loss=1.7)
    P = ref.state_dict()
    r = _call(v, q, P, labels, g_loss=1.7, flags=_lib.FLAG_BF16_PROJ)
    x = _call(v, q, P, labels, g_loss=1.7)
    l2 = lambda a, b: ((a.double().cpu() - b).norm() / b.norm()).item()   # noqa: E731
    assert _rel(r["logits"], z) < 2e-2 and l2(r["logits"], z) < 1e-2 and abs(r["loss"].item() - loss.item()) < 2e-2 * abs(loss.item())
    assert _rel(r["dv"], gv) < 2e-2 and l2(r["dv"], gv) < 1e-2
    for k in NAMES:
        assert _rel(r["d" + k], gp[k]) < 2e-2 and l2(r["d" + k], gp[k]) < 1e-2, (k, _rel(r["d" + k], gp[k]), l2(r["d" + k], gp[k]))
    assert not torch.equal(r["logits"], x["logits"]) and _rel(x["logits"], z) < TOL
    r2 = _call(v, q, P, labels, g_loss=1.7, flags=_lib.FLAG_BF16_PROJ)
    for k in r:
        if r[k] is not None:
            assert torch.equal(r[k], r2[k]), k


def test_head_logits_gradient_accumulate_and_no_input_grads():
    """The other upstream gradient (g_logits, alone and together with g_loss), accumulate = 1, dv = NULL."""
    B, d, mlp, K = 37, 64, 128, 19
    ref, v, q, labels = _case(B, d, mlp, K, seed=3)
    gx = torch.from_numpy(O.hash_normal((B, K), 77, 0.1))
    P = ref.state_dict()
    # logits only (no labels): g_logits alone
    z, _, gv, gq, gp = _oracle(ref, v, q, labels, g_loss=0.0, g_logits=gx)
    r = _call(v, q, P, None, g_logits=gx)
    assert r["loss"] is None and (r["logits"].double().cpu() - z).abs().max() < TOL
    assert _rel(r["dv"], gv) < TOL and r["dq"] is None
    for k in NAMES:
        assert _rel(r["d" + k], gp[k]) < TOL, k
    # both, added onto existing parameter gradients, no input gradients
    z, loss, gv, gq, gp = _oracle(ref, v, q, labels, g_loss=0.5, g_logits=gx)
    init = [torch.ones_like(P[k]) for k in NAMES]
    r = _call(v, q, P, labels, g_loss=0.5, g_logits=gx, accumulate=1, grads_init=init, want_dx=False)
    for k in NAMES:
        assert _rel(r["d" + k] - 1.0, gp[k]) < 2e-4, k


def test_head_is_bitwise_repeatable_and_errors_are_loud():
    from vqa_amd import _lib
    lib = _lib.load()
    B, d, mlp, K = 160, 512, 1024, 1001
    ref, v, q, labels = _case(B, d, mlp, K, seed=5)
    P = ref.state_dict()
    a = _call(v, q, P, labels, g_loss=1.0)
    for _ in range(5):
        b = _call(v, q, P, labels, g_loss=1.0)
        for k in a:
            if a[k] is not None:
                assert torch.equal(a[k], b[k]), k
    n = C.c_size_t()
    assert lib.coattn_head_workspace_bytes(0, 512, 1024, 1001, _lib.F32, C.byref(n), C.byref(n)) < 0
    assert lib.coattn_head_forward(None, None, None, None, None, None, None, 4, 8, 8, 3, _lib.F32, 0, None) < 0
    assert b"null" in lib.coattn_last_error()
    bad = labels.clone()
    bad[3] = K
    assert torch.isnan(_call(v, q, P, bad)["loss"])
    # through the module: NaN now, IndexError at the status check (nn.CrossEntropyLoss raises there)
    from vqa_amd import head as H
    ps = [P[k].float().cuda() for k in NAMES]
    H.answer_head(v.float().cuda(), q.float().cuda(), *ps, labels=labels.cuda())
    H.check_labels()
    _, l2 = H.answer_head(v.float().cuda(), q.float().cuda(), *ps, labels=bad.cuda())
    assert torch.isnan(l2)
    with pytest.raises(IndexError, match="row 3"):
        H.check_labels()


@pytest.mark.parametrize("impl", ["hip", "stock"])
def test_module_dropin(impl, monkeypatch):
    """modules.MLPClassifier: the reference's keys and list-in call; the HIP head (default on CUDA) and the stock
    modules give the same logits and gradients; lists that are rows of one [3,B,d] tensor are taken without a copy,
    unrelated tensors are stacked."""
    import vqa_amd  # noqa: F401
    from vqa_amd.head import _as_3bd
    from vqa_amd.modules import MLPClassifier
    monkeypatch.setenv("VQA_HEAD_IMPL", impl)
    B, d, mlp, K = 12, 64, 96, 11
    ref, v, q, labels = _case(B, d, mlp, K, seed=9)
    z, loss, gv, gq, gp = _oracle(ref, v, q, labels)
    mod = MLPClassifier(d, mlp, K)
    assert list(mod.state_dict().keys()) == list(ref.state_dict().keys())
    mod.load_state_dict({k: t.float() for k, t in ref.state_dict().items()})
    mod = mod.cuda()
    vg, qg = v.float().cuda().requires_grad_(True), q.float().cuda().requires_grad_(True)
    rows = [vg[l] for l in range(3)]
    assert _as_3bd(rows) is vg
    logits, l2 = mod.forward_loss(rows, [qg[l] for l in range(3)], labels.cuda())
    l2.backward()
    assert (logits.double().cpu() - z).abs().max() < TOL and abs(l2.item() - loss.item()) < TOL
    assert _rel(vg.grad, gv) < TOL and _rel(qg.grad, gq) < TOL
    for k, p in mod.named_parameters():
        assert _rel(p.grad, gp[k]) < TOL, k
    # logits-only call with three unrelated tensors (validation path: argmax of the logits)
    with torch.no_grad():
        z2 = mod([v[l].float().cuda() for l in range(3)], [q[l].float().cuda() for l in range(3)])
    assert (z2.double().cpu() - z).abs().max() < TOL
