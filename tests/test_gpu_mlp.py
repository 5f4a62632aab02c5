"""GPU: MLPClassifier + cross entropy on the HIP path (csrc/mlp.hip through the C-ABI) against the
oracle's CPU restatement of reference model.py:400-434 and nn.CrossEntropyLoss (main.py:94/:214) in float64:
logits, loss, the shared gradient with respect to v and q and the eight parameter gradients.
Tolerance (fp32, north_star 1e-4): absolute 1e-4 on logits / loss, 1e-4 of max|.| on gradients."""
import ctypes as C

import pytest
import torch

from oracle import coattn_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _case(B, d, mlp, K, seed):
    import vqa_amd  # noqa: F401
    from vqa_amd.modules import MLPClassifier
    torch.manual_seed(seed)
    mod = MLPClassifier(d, mlp, K)
    ref = O.OracleMLPClassifier(d, mlp, K).double()
    assert list(mod.state_dict().keys()) == list(ref.state_dict().keys())
    ref.load_state_dict({k: v.double() for k, v in mod.state_dict().items()})
    v = torch.from_numpy(O.hash_normal((3, B, d), seed + 1, 1.0)).float()
    q = torch.from_numpy(O.hash_normal((3, B, d), seed + 2, 0.5)).float()
    labels = torch.from_numpy((O.hash_uniform(B, seed + 3) * K).astype("int64")).clamp_(0, K - 1)
    return mod, ref, v, q, labels


def _rel(a, b):
    return ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("shape", [(160, 512, 1024, 1001), (5, 64, 96, 7), (32, 128, 256, 3), (3, 20, 12, 5)],
                         ids=lambda s: "B%d_d%d_mlp%d_K%d" % s)
def test_mlp_and_cross_entropy_vs_oracle(shape, monkeypatch):
    from vqa_amd.mlp import CrossEntropyLoss
    monkeypatch.setenv("VQA_MLP_IMPL", "hip")
    B, d, mlp, K = shape
    mod, ref, v, q, labels = _case(B, d, mlp, K, 40 + d)
    vr, qr = v.double().requires_grad_(True), q.double().requires_grad_(True)
    zr = ref([vr[l] for l in range(3)], [qr[l] for l in range(3)])
    lr = torch.nn.functional.cross_entropy(zr, labels)
    lr.backward()
    mod = mod.cuda()
    vg, qg = v.cuda().requires_grad_(True), q.cuda().requires_grad_(True)
    z = mod([vg[l] for l in range(3)], [qg[l] for l in range(3)])
    loss = CrossEntropyLoss()(z, labels.cuda())
    loss.backward()
    assert z.shape == (B, K)
    assert (z.double().cpu() - zr.detach()).abs().max().item() < TOL
    assert abs(loss.item() - lr.item()) < TOL
    assert _rel(vg.grad, vr.grad) < TOL and _rel(qg.grad, qr.grad) < TOL
    for (k, p), (_, pr) in zip(mod.named_parameters(), ref.named_parameters()):
        assert _rel(p.grad, pr.grad) < TOL, k


def test_module_surface_matches_stock_modules(monkeypatch):
    """Same nn.Module, HIP path vs its own stock-torch path on the GPU: values and gradients; the co-attention's
    [3,B,d] buffers are consumed without a copy and receive their gradient directly."""
    from vqa_amd.mlp import as_level_stack
    mod, _, v, q, labels = _case(16, 128, 256, 11, 5)
    mod = mod.cuda()
    outs = {}
    for impl in ("hip", "stock"):
        monkeypatch.setenv("VQA_MLP_IMPL", impl)
        mod.zero_grad()
        vb, qb = v.cuda().requires_grad_(True), q.cuda().requires_grad_(True)
        views_v, views_q = [vb[l] for l in range(3)], [qb[l] for l in range(3)]
        if impl == "hip":
            assert as_level_stack(views_v) is vb and as_level_stack(views_q) is qb
            assert as_level_stack([vb[1], vb[0], vb[2]]) is not vb
        z = mod(views_v, views_q)
        torch.nn.functional.cross_entropy(z, labels.cuda()).backward()
        outs[impl] = [z.detach(), vb.grad, qb.grad] + [p.grad.clone() for p in mod.parameters()]
    for a, b in zip(outs["hip"], outs["stock"]):
        assert (a - b).abs().max() <= 1e-5 * max(1.0, b.abs().max().item())


def test_cross_entropy_semantics():
    """Mean reduction, upstream gradient scaling, no-grad / inference call, labels out of range -> NaN."""
    from vqa_amd.mlp import cross_entropy
    B, K = 37, 1001
    z = torch.from_numpy(O.hash_normal((B, K), 3, 3.0)).float().cuda()
    lab = torch.from_numpy((O.hash_uniform(B, 4) * K).astype("int64")).clamp_(0, K - 1).cuda()
    zr = z.double().requires_grad_(True)
    (2.5 * torch.nn.functional.cross_entropy(zr, lab)).backward()
    zg = z.clone().requires_grad_(True)
    (2.5 * cross_entropy(zg, lab)).backward()
    assert (zg.grad.double() - zr.grad).abs().max() < 1e-6
    with torch.no_grad():
        l0 = cross_entropy(z, lab)
    assert abs(l0.item() - torch.nn.functional.cross_entropy(z.double(), lab).item()) < 1e-5
    bad = lab.clone()
    bad[3] = K
    assert torch.isnan(cross_entropy(z, bad))
    # rows with a huge logit: log-sum-exp stays finite
    z2 = z.clone()
    z2[0, 5] = 8.0e4
    assert torch.isfinite(cross_entropy(z2, lab))


def test_c_abi_errors_and_inference(monkeypatch):
    """Loud argument errors; inference (saved = NULL) equals the training-mode forward."""
    from vqa_amd import _lib
    monkeypatch.setenv("VQA_MLP_IMPL", "hip")
    lib = _lib.load()
    assert lib.coattn_mlp_forward(None, None, None, None, None, None, 4, 64, 64, 3, _lib.F32, 0, None) < 0
    assert b"null" in lib.coattn_last_error()
    n = C.c_size_t()
    assert lib.coattn_ce_workspace_bytes(0, 5, _lib.F32, C.byref(n)) < 0
    assert lib.coattn_mlp_workspace_bytes(4, 64, 64, 3, 7, None, None, None) < 0
    mod, _, v, q, _ = _case(9, 64, 128, 6, 2)
    mod = mod.cuda()
    views = lambda t: [t[l] for l in range(3)]
    with torch.no_grad():
        z0 = mod(views(v.cuda()), views(q.cuda()))
    z1 = mod(views(v.cuda().requires_grad_(True)), views(q.cuda()))
    assert torch.equal(z0, z1.detach())
    with pytest.raises(RuntimeError):
        from vqa_amd.mlp import mlp_classify
        mlp_classify(v, q, *[p for p in mod.parameters()])          # CPU tensors: no fallback inside the op
