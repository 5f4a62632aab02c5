"""GPU: PhraseConvPool on the HIP path (csrc/phrase.hip through the C-ABI) against the oracle's CPU
restatement of reference model.py:301-334 in float64 -- forward, input gradient and the six
parameter gradients; ragged zero-padded questions, odd sizes (generic GEMM path), inference mode."""
import pytest
import torch

from oracle import coattn_oracle as O
from oracle import net_oracle as NO

pytestmark = pytest.mark.gpu


def _case(B, T, E, seed):
    import vqa_amd  # noqa: F401
    from vqa_amd.modules import PhraseConvPool
    torch.manual_seed(seed)
    mod = PhraseConvPool(E)
    ref = NO.OraclePhraseConvPool(E).double()
    ref.load_state_dict({k: v.double() for k, v in mod.state_dict().items()})
    x = torch.from_numpy(O.hash_normal((B, T, E), seed + 1, 1.0)).float()
    for b in range(B):
        x[b, max(1, T - 3 * b):] = 0                          # ragged: zero rows past each length
    g = torch.from_numpy(O.hash_normal((B, T, E), seed + 2, 1.0)).float()
    return mod, ref, x, g


# (12, 26, 512) and (40, 13, 384): >= 128 rows and 128-aligned channels -> the hand-scheduled GEMMs (k bands with 4 / 3
# column tiles per band, masked split-K weight gradient); (4, 26, 512): weight gradient only; the rest: general GEMM
@pytest.mark.parametrize("fast", [True, False], ids=["tolerance_mode", "exact_mode"])
@pytest.mark.parametrize("shape", [(3, 26, 64), (2, 5, 20), (1, 1, 4), (4, 26, 512), (2, 7, 36), (12, 26, 512), (40, 13, 384)],
                         ids=lambda s: "B%d_T%d_E%d" % s)
def test_phrase_conv_pool_vs_oracle(shape, fast):
    """fast: the tolerance mode train.Trainer sets (Z = Xcat Wcat^T on two FP16 pieces, the gradient products on two bf16
    pieces; include/coattn.h COATTN_FLAG_FAST16), held to 5e-5; else the module's default, fp32-accurate products, 1e-5."""
    B, T, E = shape
    mod, ref, x, g = _case(B, T, E, 11 + E)
    mod.fast_products = fast
    tol = 5e-5 if fast else 1e-5
    assert list(mod.state_dict().keys()) == list(ref.state_dict().keys())
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    yr.backward(g.double())
    mod = mod.cuda()
    xg = x.cuda().requires_grad_(True)
    y = mod(xg)
    y.backward(g.cuda())
    assert y.shape == (B, T, E)
    err = (y.detach().cpu().double() - yr.detach()).abs().max().item()
    assert err < tol, err
    def rel(a, b):
        return (a.cpu().double() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
    worst = {"dx": rel(xg.grad, xr.grad)}
    for (k, p), (_, pr) in zip(mod.named_parameters(), ref.named_parameters()):
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        worst[k] = rel(p.grad, pr.grad)
    print("phrase", shape, "fast" if fast else "exact", "fwd %.1e" % err, {k: "%.1e" % e for k, e in worst.items()})
    assert max(worst.values()) < tol, worst
    import vqa_amd
    vqa_amd.check_range()                                     # (tolerance mode: nothing left the FP16-piece range)


def test_phrase_range_report():
    """A conv weight beyond the scaled FP16 weight image's range (|W| > 255.87): the tolerance mode reports it
    (vqa_amd.check_range raises RangeError); the exact mode computes the reference's value."""
    import vqa_amd
    mod, ref, x, g = _case(12, 26, 512, 77)
    with torch.no_grad():
        mod.conv_bigram[1].weight[5, 9, 1] = 400.0
    ref.load_state_dict({k: v.double() for k, v in mod.state_dict().items()})
    yr = ref(x.double())
    mod = mod.cuda()
    mod.fast_products = True
    xg = x.cuda().requires_grad_(True)
    mod(xg)
    with pytest.raises(vqa_amd.RangeError):
        vqa_amd.check_range()
    vqa_amd.check_range()                                     # (the report was consumed by the check that raised)
    mod.fast_products = False
    y = mod(xg)
    vqa_amd.check_range()
    assert (y.detach().cpu().double() - yr).abs().max().item() < 1e-5


def test_phrase_matches_stock_modules_on_gpu_and_inference(monkeypatch):
    """Same module object: HIP path == its own stock torch path (MIOpen convs) to fp32 tolerance;
    no_grad forward keeps nothing."""
    mod, _, x, _ = _case(5, 26, 256, 3)
    mod = mod.cuda()
    xg = x.cuda()
    with torch.no_grad():
        y_hip = mod(xg)
    monkeypatch.setenv("VQA_PHRASE_IMPL", "stock")
    with torch.no_grad():
        y_stock = mod(xg)
    assert (y_hip - y_stock).abs().max().item() < 1e-4


def test_phrase_errors_are_loud():
    from vqa_amd.phrase import phrase_conv_pool
    E = 8
    x = torch.zeros(2, 3, E)
    W = [torch.zeros(E, E, k) for k in (1, 2, 3)]
    b = torch.zeros(E)
    with pytest.raises(RuntimeError, match="GPU"):
        phrase_conv_pool(x, W[0], b, W[1], b, W[2], b)
    with pytest.raises(RuntimeError, match="weights"):
        phrase_conv_pool(x.cuda(), W[1].cuda(), b.cuda(), W[1].cuda(), b.cuda(), W[2].cuda(), b.cuda())


@pytest.mark.parametrize("B,T,E", [(6, 26, 256), (16, 26, 256), (32, 26, 512)], ids=lambda v: str(v))
def test_phrase_bf16_mfma_mode(B, T, E):
    """bf16=True (what CUDA autocast selects): bf16-rounded operands on the bf16 MFMA, fp32 accumulation
    -> values and gradients within bf16 tolerance of the float64 oracle, not identical to the fp32 mode.
    B = 6: gemm_w / gemm_tn in single-piece mode (B T rows: not a multiple of 256 / 32); B = 16, 32: the wide-shape
    kernels of gemm_bf.hip for Z, dXcat (k bands) and dWcat (masked tap blocks)."""
    from vqa_amd.phrase import phrase_conv_pool
    mod, ref, x, g = _case(B, T, E, 21)
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    yr.backward(g.double())
    mod = mod.cuda()
    u, b, t = mod.conv_unigram[1], mod.conv_bigram[1], mod.conv_trigram[1]
    xg = x.cuda().requires_grad_(True)
    y = phrase_conv_pool(xg, u.weight, u.bias, b.weight, b.bias, t.weight, t.bias, bf16=True)
    y.backward(g.cuda())
    y32 = phrase_conv_pool(xg.detach(), u.weight, u.bias, b.weight, b.bias, t.weight, t.bias, bf16=False)
    assert (y.detach() - y32).abs().max().item() > 0
    assert (y.detach().cpu().double() - yr.detach()).abs().max().item() < 3e-2
    def rel(a, r):      # relative L2 error: a reduced-precision max-pool may route single gradients to another channel
        return ((a.cpu().double() - r).norm() / r.norm()).item()
    assert rel(xg.grad, xr.grad) < 1e-1
    for (k, p), (_, pr) in zip(mod.named_parameters(), ref.named_parameters()):
        assert rel(p.grad, pr.grad) < 1e-1, (k, rel(p.grad, pr.grad))
    with torch.autocast("cuda", dtype=torch.bfloat16):          # autocast picks the bf16 mode, output stays fp32
        ya = mod(xg.detach())
    assert ya.dtype == torch.float32 and torch.equal(ya, y.detach())
