"""GPU: the fused cross entropy (csrc/ce.hip through the C-ABI) against nn.CrossEntropyLoss semantics
(main.py:94 / :214) in float64, and the stock MLPClassifier of modules.py against the oracle's restatement of
reference model.py:400-434 (same state_dict keys, same values).  Tolerance (fp32, north_star 1e-4)."""
import ctypes as C

import pytest
import torch

from oracle import coattn_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.mark.parametrize("shape", [(160, 512, 1024, 1001), (5, 64, 96, 7), (3, 20, 12, 5)],
                         ids=lambda s: "B%d_d%d_mlp%d_K%d" % s)
def test_mlp_head_and_cross_entropy_vs_oracle(shape):
    """logits of the stock head + the HIP cross entropy, loss and every gradient against the float64 oracle."""
    import vqa_amd  # noqa: F401
    from vqa_amd.loss import cross_entropy
    from vqa_amd.modules import MLPClassifier
    B, d, mlp, K = shape
    torch.manual_seed(B + d)
    mod = MLPClassifier(d, mlp, K)
    ref = O.OracleMLPClassifier(d, mlp, K).double()
    assert list(mod.state_dict().keys()) == list(ref.state_dict().keys())
    ref.load_state_dict({k: v.double() for k, v in mod.state_dict().items()})
    v = torch.from_numpy(O.hash_normal((3, B, d), 11, 1.0)).float()
    q = torch.from_numpy(O.hash_normal((3, B, d), 12, 0.5)).float()
    labels = torch.from_numpy((O.hash_uniform(B, 13) * K).astype("int64")).clamp_(0, K - 1)
    vr, qr = v.double().requires_grad_(True), q.double().requires_grad_(True)
    zr = ref([vr[l] for l in range(3)], [qr[l] for l in range(3)])
    lr = torch.nn.functional.cross_entropy(zr, labels)
    lr.backward()
    mod = mod.cuda()
    vg, qg = v.cuda().requires_grad_(True), q.cuda().requires_grad_(True)
    zg = mod([vg[l] for l in range(3)], [qg[l] for l in range(3)])
    lg = cross_entropy(zg, labels.cuda())
    lg.backward()
    assert (zg.detach().double().cpu() - zr.detach()).abs().max() < TOL and abs(lg.item() - lr.item()) < TOL

    def rel(a, b):
        return ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()

    assert rel(vg.grad, vr.grad) < TOL and rel(qg.grad, qr.grad) < TOL
    for (k, pg), (_, pr) in zip(mod.named_parameters(), ref.named_parameters()):
        assert rel(pg.grad, pr.grad) < TOL, k


def test_cross_entropy_semantics():
    """Mean reduction, upstream gradient scaling, no-grad / inference call, labels out of range -> NaN + IndexError at
    the status check (nn.CrossEntropyLoss raises)."""
    from vqa_amd.loss import cross_entropy
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
    from vqa_amd import loss as L
    L.check_labels()                                   # all labels in range so far: no error
    bad = lab.clone()
    bad[3] = K
    assert torch.isnan(cross_entropy(z, bad))          # asynchronous: NaN now, the error at the next check
    with pytest.raises(IndexError, match="row 3"):
        L.check_labels()
    cross_entropy(z, lab)
    L.check_labels()                                   # the status word is cleared by every call
    # rows with a huge logit: log-sum-exp stays finite
    z2 = z.clone()
    z2[0, 5] = 8.0e4
    assert torch.isfinite(cross_entropy(z2, lab))


@pytest.mark.parametrize("B", [1, 2, 63, 160, 257, 1000])
def test_cross_entropy_mean_is_one_launch_and_order_independent(B):
    """The mean is added by whichever workgroup finishes last (ce.hip: an integer ticket): the value must not depend on which
    one that is -- repeated calls bit for bit, equal to the fixed-order sum of the row losses (strided per thread of 256,
    wave sums, (0 + 1) + (2 + 3)) within fp32 rounding of the float64 mean -- and the ticket word must be back at 0."""
    from vqa_amd import _lib
    from vqa_amd.loss import cross_entropy
    K = 1001
    z = torch.from_numpy(O.hash_normal((B, K), 11, 2.0)).float().cuda()
    lab = torch.from_numpy((O.hash_uniform(B, 12) * K).astype("int64")).clamp_(0, K - 1).cuda()
    ref = torch.nn.functional.cross_entropy(z.double(), lab).item()
    first = cross_entropy(z, lab)
    assert abs(first.item() - ref) < 2e-6 * max(1.0, abs(ref))
    for _ in range(20):
        assert torch.equal(cross_entropy(z, lab), first)
    # through the C-ABI: the workspace's ticket word is 0 again after every call
    lib = _lib.load()
    n = C.c_size_t()
    assert lib.coattn_ce_workspace_bytes(B, K, _lib.F32, C.byref(n)) == 0
    ws = torch.full((n.value // 4,), float("nan"), device="cuda")          # (uninitialised by contract: the call clears its words)
    loss = torch.empty((), device="cuda"); dz = torch.empty(B, K, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert lib.coattn_ce_forward(z.data_ptr(), lab.data_ptr(), loss.data_ptr(), dz.data_ptr(), ws.data_ptr(), B, K, _lib.F32, st) == 0
        assert torch.equal(loss, first)
    words = ws.view(torch.int32)[(B + 63) // 64 * 64:][:2].tolist()
    assert words == [0, 0], words
    assert lib.coattn_ce_status(ws.data_ptr(), B, st) == 0


def test_c_abi_errors():
    """Loud argument errors of the cross-entropy entry points."""
    from vqa_amd import _lib
    lib = _lib.load()
    n = C.c_size_t()
    assert lib.coattn_ce_workspace_bytes(0, 5, _lib.F32, C.byref(n)) < 0
    assert lib.coattn_ce_forward(None, None, None, None, None, 4, 7, _lib.F32, None) < 0
    assert b"null" in lib.coattn_last_error()
    assert not hasattr(lib, "coattn_mlp_forward")          # the HIP head was removed: stock modules
