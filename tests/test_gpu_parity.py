"""GPU parity (the tests proper): HIP path through the C-ABI vs the golden vectors produced by
the imported reference (tests/golden, float64 run) and vs the CPU oracle on the same inputs.

Tolerances (fp32, BASELINE.json north_star: 1e-4): forward outputs / attention maps absolute
1e-4 (observed ~1e-6); gradients 1e-4 relative to the tensor's max magnitude."""
import pytest
import torch

from oracle import coattn_oracle as O
from tests import _golden as G

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-4
GRAD_TOL = 1e-4
# On the reference's goldens the asserts are tighter than the contract, so that green itself certifies margin:
# the tolerance mode (COATTN_FLAG_FAST16: what train.Trainer runs, and what run_hip passes unless exact3) is held to 5e-5
# (observed: H_q 2.0e-5, gradients 1.5e-5 of max|.|), the exact mode (flags = 0) to 1e-5.
GOLDEN_TOL_FAST = 5e-5
GOLDEN_TOL_EXACT = 1e-5


def _impls(name):
    import vqa_amd
    c = G.CASES[name]
    out = ["general"]
    if vqa_amd._lib.load().coattn_fused_supported(c["B"], c["N"], c["T"], c["d"], 3, 0):
        out.append("fused")
    return out


@pytest.mark.parametrize("name", sorted(G.CASES))
def test_forward_backward_vs_reference_golden(name):
    from tests._hip import run_hip
    gold = G.load(name)
    V, Qs, P, gv, gq = G.build_case(name, torch.float32)
    from tests._hip import LAYOUTS
    for impl in _impls(name):
        for layout in LAYOUTS:      # channel-major [B,d,N] (NCHW encoder) and location-major [B,N,d] (channels_last)
            r = run_hip(V, Qs, P, gv, gq, impl=impl, layout=layout)
            for k, t in r.items():
                assert torch.isfinite(t).all(), (impl, layout, k)
            fe = G.fwd_errors(r, gold, "64")
            ge = G.grad_errors(r, gold, "64")
            print(name, impl, layout, "fwd", {k: "%.1e" % v for k, v in fe.items()},
                  "grad", {k: "%.1e" % v for k, v in ge.items()})
            assert max(fe.values()) < GOLDEN_TOL_FAST, (impl, layout, fe)
            assert max(ge.values()) < GOLDEN_TOL_FAST, (impl, layout, ge)
            from tests import _hip
            assert _hip.last_status[0] == 0, _hip.last_status        # no operand left the FP16-piece range


@pytest.mark.parametrize("name", sorted(G.CASES))
def test_exact_split_flag_vs_reference_golden(name):
    """flags = 0, the C-ABI's default (= COATTN_FLAG_EXACT3): every contraction on the three-piece split -- the goldens hold
    at fp32 rounding level, in both physical layouts."""
    from tests._hip import run_hip
    if "fused" not in _impls(name):
        pytest.skip("general-shape path: always exact")
    gold = G.load(name)
    V, Qs, P, gv, gq = G.build_case(name, torch.float32)
    for layout in ("cm", "lm"):
        r = run_hip(V, Qs, P, gv, gq, impl="fused", layout=layout, exact3=True)
        fe, ge = G.fwd_errors(r, gold, "64"), G.grad_errors(r, gold, "64")
        print(name, layout, "exact fwd %.1e grad %.1e" % (max(fe.values()), max(ge.values())))
        assert max(fe.values()) < GOLDEN_TOL_EXACT and max(ge.values()) < GOLDEN_TOL_EXACT, (layout, fe, ge)
    # the default widths differ from it only inside their budget (tests/test_split_emulation.py)
    r2 = run_hip(V, Qs, P, gv, gq, impl="fused", layout="lm")
    for k in ("dQ", "dW_v.weight", "dW_q.weight"):
        d = (r2[k] - r[k]).abs().max() / r[k].abs().max()
        assert d < 5e-5, (k, float(d))


@pytest.mark.parametrize("impl", ["general", "fused"])
def test_module_dropin_autograd(impl, monkeypatch):
    """nn.Module surface (model.py:337-397): same state_dict keys, list-in/list-out, autograd; W_b dead."""
    import vqa_amd
    B, N, T, d = 4, 49, 26, 256
    if impl == "fused" and not vqa_amd._lib.load().coattn_fused_supported(B, N, T, d, 3, 0):
        pytest.skip("no fused configuration for this shape")
    monkeypatch.setenv("COATTN_IMPL", impl)
    torch.manual_seed(0)
    ref = O.OracleParallelCoAttention(d, as_executed=True)
    mod = vqa_amd.ParallelCoAttention(d)
    assert list(mod.state_dict().keys()) == list(ref.state_dict().keys())
    mod.load_state_dict(ref.state_dict())
    mod = mod.cuda()
    V, Qs = O.make_inputs(B, N, T, d, 9, lens=[26, 20, 5, 1], scale_q=(2.0 / d) ** 0.5)
    x_img = V.permute(0, 2, 1)                              # the encoder's permuted view
    x_ref = x_img.clone().requires_grad_(True)
    Qr = [q.clone().requires_grad_(True) for q in Qs]
    vs, qs = ref(x_ref, Qr)
    x_gpu = V.cuda().permute(0, 2, 1).requires_grad_(True)
    Qg = [q.cuda().requires_grad_(True) for q in Qs]
    vg, qg = mod(x_gpu, Qg)
    assert isinstance(vg, list) and len(vg) == 3 and vg[0].shape == (B, d)
    w = [torch.randn(B, d) for _ in range(6)]
    (sum((vs[l] * w[l]).sum() + (qs[l] * w[3 + l]).sum() for l in range(3))).backward()
    (sum((vg[l] * w[l].cuda()).sum() + (qg[l] * w[3 + l].cuda()).sum() for l in range(3))).backward()
    for l in range(3):
        assert (vg[l].cpu() - vs[l]).abs().max() < FWD_TOL and (qg[l].cpu() - qs[l]).abs().max() < FWD_TOL

    def rel(a, b):
        return ((a.cpu() - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()

    assert rel(x_gpu.grad, x_ref.grad) < GRAD_TOL
    for l in range(3):
        assert rel(Qg[l].grad, Qr[l].grad) < GRAD_TOL
    for (k, pg), (_, pr) in zip(mod.named_parameters(), ref.named_parameters()):
        if k.startswith("W_b"):
            assert pg.grad is None and pr.grad is None           # dead layer (model.py:347 vs :377)
        elif k in ("w_v.bias", "w_q.bias"):
            assert (pg.grad.cpu() - pr.grad).abs().max() < 1e-4  # analytically zero
        else:
            assert rel(pg.grad, pr.grad) < GRAD_TOL, k


@pytest.mark.parametrize("exact3", [False, True], ids=["fast16", "exact"])
@pytest.mark.parametrize("impl", ["general", "fused"])
def test_frozen_image_features_and_accumulate(impl, exact3):
    """dV = NULL (frozen encoder, model.py:239-241) leaves the other gradients unchanged, and
    accumulate=1 adds into the parameter gradients."""
    import vqa_amd
    from tests._hip import run_hip
    name = "g5_d256"
    c = G.CASES[name]
    if impl == "fused" and not vqa_amd._lib.load().coattn_fused_supported(c["B"], c["N"], c["T"], c["d"], 3, 0):
        pytest.skip("no fused configuration for this shape")
    if impl == "general" and not exact3:
        pytest.skip("the general-shape path has one arithmetic (exact): run once")
    V, Qs, P, gv, gq = G.build_case(name, torch.float32)
    a = run_hip(V, Qs, P, gv, gq, impl=impl, exact3=exact3)
    b = run_hip(V, Qs, P, gv, gq, impl=impl, need_dv=False, exact3=exact3)
    for k in G.GRAD_KEYS:
        if k != "dV_phys":          # same values; summation order over the levels may differ
            assert (a[k] - b[k]).abs().max() <= 1e-5 * max(1e-3, a[k].abs().max().item()), k
    names = ("W_v.weight", "W_v.bias", "W_q.weight", "W_q.bias", "w_v.weight", "w_v.bias", "w_q.weight", "w_q.bias")
    init = [torch.ones_like(P[k]) for k in names]
    cacc = run_hip(V, Qs, P, gv, gq, impl=impl, accumulate=1, grads_init=init, exact3=exact3)
    for k in names:
        assert (cacc["d" + k] - (a["d" + k] + 1.0)).abs().max() < 1e-4 * max(1.0, a["d" + k].abs().max().item()), k


_FULL64 = {}


def _full_batch_grads_float64(N, V, Qs, P, gv, gq):
    """Parameter gradients of the FULL batch from the oracle run in float64 on the fp32 inputs (cached per grid: the six
    parametrisations below share two sets of inputs)."""
    if N not in _FULL64:
        d64 = lambda t: t.double()                                    # noqa: E731
        g = O.coattn_backward(d64(V), [d64(q) for q in Qs], {k: d64(v) for k, v in P.items()}, d64(gv), d64(gq))
        _FULL64[N] = {k: g["d" + k] for k in O.PARAM_KEYS}
    return _FULL64[N]


@pytest.mark.parametrize("exact3", [False, True], ids=["fast16", "exact"])
@pytest.mark.parametrize("impl,layout,N", [("general", "cm", 196), ("fused", "cm", 196), ("fused", "lm", 196),
                                           ("general", "lm", 49), ("fused", "cm", 49), ("fused", "lm", 49)])
def test_full_size_cfg2_properties(impl, layout, N, exact3):
    """BASELINE config 2 (B=160, T=26, d=512) at the reference's own grid (448x448 -> N=196) and at the grid of
    BASELINE's 224x224 images (N=49), both physical layouts of the image features: oracle on a sample subset +
    size-independent properties (attention maps are distributions; v inside the range of V; per-sample
    independence; backward linear in the upstream gradient) + full-batch parameter gradients vs the oracle."""
    import vqa_amd
    from tests._hip import run_hip
    B, T, d = 160, 26, 512
    if impl != "general" and not vqa_amd._lib.load().coattn_fused_supported(B, N, T, d, 3, 0):
        pytest.skip("no fused configuration for this shape")
    if impl == "general" and not exact3:
        pytest.skip("the general-shape path has one arithmetic (exact): run once")
    # exact (flags = 0, the reference's arithmetic): held to 2e-5 where the tolerance mode is held to the contract's 1e-4
    fwd_tol, grad_tol = (2e-5, 2e-5) if exact3 else (FWD_TOL, GRAD_TOL)
    lens = sorted([26] + [3 + (7 * i) % 24 for i in range(B - 1)], reverse=True)
    P = O.make_params(d, 5)
    V, Qs = O.make_inputs(B, N, T, d, 77, lens=lens, scale_q=(2.0 / d) ** 0.5)
    gv = torch.from_numpy(O.hash_normal((3, B, d), 901)).float()
    gq = torch.from_numpy(O.hash_normal((3, B, d), 902)).float()
    r = run_hip(V, Qs, P, gv, gq, impl=impl, layout=layout, exact3=exact3)
    assert (r["a_v"].sum(-1) - 1).abs().max() < 1e-5 and (r["a_q"].sum(-1) - 1).abs().max() < 1e-5
    Vd = V.cuda()
    assert (r["v"] <= Vd.max(2).values[None] + 1e-5).all() and (r["v"] >= Vd.min(2).values[None] - 1e-5).all()
    # oracle on 3 samples of the batch (forward is per-sample independent)
    idx = [0, 77, 159]
    d64 = lambda t: t.double()                                        # noqa: E731
    P64 = {k: d64(v) for k, v in P.items()}
    f = O.coattn_forward(d64(V[idx]), [d64(q[idx]) for q in Qs], P64)
    ef = max(float((r["v"][:, idx].cpu().double() - f["v"]).abs().max()), float((r["q"][:, idx].cpu().double() - f["q"]).abs().max()))
    g = O.coattn_backward(d64(V[idx]), [d64(q[idx]) for q in Qs], P64, d64(gv[:, idx]), d64(gq[:, idx]))
    scale = g["dV_phys"].abs().max()
    eg = max(float((r["dV_phys"][idx].cpu().double() - g["dV_phys"]).abs().max() / scale),
             float((r["dQ"][:, idx].cpu().double() - g["dQ"]).abs().max() / g["dQ"].abs().max()))
    print("full batch N=%d %s %s %s: sample subset vs float64 oracle: fwd %.1e, dV/dQ %.1e" % (N, impl, layout, "exact" if exact3 else "fast16", ef, eg))
    assert ef < fwd_tol and eg < grad_tol, (ef, eg)
    # linearity: backward(2*g) == 2*backward(g)
    r2 = run_hip(V, Qs, P, 2 * gv, 2 * gq, impl=impl, layout=layout, exact3=exact3)
    for k in ("dV_phys", "dQ", "dW_v.weight", "dW_q.weight", "dw_v.weight"):
        assert (r2[k] - 2 * r[k]).abs().max() <= 2e-5 * max(1e-3, r[k].abs().max().item()), k
    # full-batch parameter gradients (sums over 160 x N x 3 rows) vs the oracle in FLOAT64 at the contract's 1e-4 -- the
    # tolerance mode at BASELINE's own size (this used to be 2e-4 against an fp32 CPU oracle)
    gf = _full_batch_grads_float64(N, V, Qs, P, gv, gq)
    worst = {}
    for k in O.PARAM_KEYS:
        if k.endswith("w_v.bias") or k.endswith("w_q.bias"):
            continue
        ref = gf[k]
        worst[k] = float((r["d" + k].cpu().double() - ref).abs().max() / ref.abs().max())
    print("full batch N=%d %s %s %s: parameter-gradient errors vs float64" % (N, impl, layout, "exact" if exact3 else "fast16"),
          {k: "%.1e" % e for k, e in worst.items()})
    assert max(worst.values()) < grad_tol, worst
