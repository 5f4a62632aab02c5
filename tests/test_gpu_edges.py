"""GPU: edge cases of the drop-in surface -- inference mode, input layouts, autocast, determinism,
shape limits of the fused kernels (auto dispatch), error behaviour."""
import pytest
import torch

from oracle import coattn_oracle as O

pytestmark = pytest.mark.gpu


def _mod(d, seed=3):
    import vqa_amd
    m = vqa_amd.ParallelCoAttention(d)
    m.load_state_dict(O.make_params(d, seed))
    return m.cuda()


def test_inference_equals_training_forward_and_layouts():
    B, N, T, d = 5, 49, 26, 512
    m = _mod(d)
    V, Qs = O.make_inputs(B, N, T, d, 31, lens=[26, 20, 9, 2, 1], scale_q=(2.0 / d) ** 0.5)
    Vg = V.cuda(); Qg = [q.cuda() for q in Qs]
    view = Vg.permute(0, 2, 1)                                  # encoder's permuted view (no copy)
    v_tr, q_tr = m(view.clone().permute(0, 2, 1).contiguous().permute(0, 2, 1).requires_grad_(True), Qg)
    with torch.no_grad():
        v_inf, q_inf = m(view, Qg)                              # saved = NULL path
        v_cl, q_cl = m(view.contiguous(), Qg)                   # [B,N,d]-contiguous (channels_last encoder)
    f = O.coattn_forward(V, Qs, O.make_params(d, 3))
    for l in range(3):
        assert torch.equal(v_tr[l], v_inf[l]) and torch.equal(q_tr[l], q_inf[l])
        # the location-major buffer runs its own kernels: same values to rounding, not bitwise
        assert (v_cl[l] - v_inf[l]).abs().max() < 1e-5 and (q_cl[l] - q_inf[l]).abs().max() < 1e-5
    assert (torch.stack(v_inf).cpu() - f["v"]).abs().max() < 1e-4
    assert (torch.stack(v_cl).cpu() - f["v"]).abs().max() < 1e-4


@pytest.mark.parametrize("layout", ["lm", "cm", "cm49", "strided"])
def test_module_takes_the_features_where_they_lie(layout):
    """No copy between encoder and kernels for the two native layouts (the same storage is read; the gradient comes
    back in the layout of x_img), a copy only for anything else -- that includes channel-major rows that are not
    16-byte multiples (N = 49); values and gradients equal the oracle's."""
    B, N, T, d = 4, (52 if layout == "cm" else 49), 26, 512
    m = _mod(d)
    V, Qs = O.make_inputs(B, N, T, d, 33, lens=[26, 17, 4, 1], scale_q=(2.0 / d) ** 0.5)
    P = O.make_params(d, 3)
    gv = torch.from_numpy(O.hash_normal((3, B, d), 11)).float()
    gq = torch.from_numpy(O.hash_normal((3, B, d), 12)).float()
    if layout == "lm":
        x = V.cuda().permute(0, 2, 1).contiguous()                    # [B,N,d] contiguous
    elif layout in ("cm", "cm49"):
        x = V.cuda().permute(0, 2, 1)                                 # permuted view of [B,d,N]
    else:
        x = torch.zeros(B, N, 2 * d, device="cuda")[:, :, ::2]        # neither: stride 2 along d
        x.copy_(V.cuda().permute(0, 2, 1))
    x.requires_grad_(True)
    import sys
    import vqa_amd  # noqa: F401
    native = sys.modules["vqa_amd.coattention"]._native_layout(x)
    assert (native.data_ptr() == x.data_ptr()) == (layout in ("lm", "cm"))
    Qg = [q.cuda().requires_grad_(True) for q in Qs]
    v, q = m(x, Qg)
    (sum((v[l] * gv[l].cuda()).sum() + (q[l] * gq[l].cuda()).sum() for l in range(3))).backward()
    f = O.coattn_forward(V, Qs, P)
    g = O.coattn_backward(V, Qs, P, gv, gq)
    assert (torch.stack(v).cpu() - f["v"]).abs().max() < 1e-4 and (torch.stack(q).cpu() - f["q"]).abs().max() < 1e-4
    assert x.grad.shape == x.shape
    if layout in ("lm", "cm"):
        assert x.grad.stride() == x.stride()
    ref = g["dV_phys"].permute(0, 2, 1)
    assert (x.grad.cpu() - ref).abs().max() <= 1e-4 * ref.abs().max()
    for l in range(3):
        assert (Qg[l].grad.cpu() - g["dQ"][l]).abs().max() <= 1e-4 * g["dQ"].abs().max()


def test_autocast_keeps_the_op_in_fp32():
    B, N, T, d = 3, 49, 26, 256
    m = _mod(d)
    V, Qs = O.make_inputs(B, N, T, d, 32, lens=[26, 5, 1], scale_q=(2.0 / d) ** 0.5)
    x = V.cuda().permute(0, 2, 1)
    Qg = [q.cuda().requires_grad_(True) for q in Qs]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        v, q = m(x.bfloat16(), [t.bfloat16() for t in Qg])      # bf16 features from an autocast encoder
    assert v[0].dtype == torch.float32
    sum(t.sum() for t in v + q).backward()
    assert all(t.grad is not None and torch.isfinite(t.grad).all() for t in Qg)
    f = O.coattn_forward(V.bfloat16().float(), [t.bfloat16().float() for t in Qs], O.make_params(d, 3))
    assert (torch.stack(v).cpu() - f["v"]).abs().max() < 1e-3


def test_bitwise_deterministic():
    from tests._hip import run_hip
    from tests import _golden as G
    V, Qs, P, gv, gq = G.build_case("g5_cfg2_scaled", torch.float32)
    for layout, exact3 in [(lay, m) for lay in ("cm", "lm") for m in (False, True)]:
        a = run_hip(V, Qs, P, gv, gq, impl="auto", layout=layout, exact3=exact3)
        b = run_hip(V, Qs, P, gv, gq, impl="auto", layout=layout, exact3=exact3)
        for k in a:
            assert torch.equal(a[k], b[k]), (layout, exact3, k)


@pytest.mark.parametrize("B,N,T,d", [(1, 1, 1, 32), (2, 208, 28, 512), (2, 209, 26, 512), (2, 196, 29, 512),
                                     (3, 64, 7, 256), (2, 65, 26, 512), (1, 16, 16, 128),
                                     (2, 49, 26, 2048), (2, 196, 26, 1024), (2, 49, 20, 768)])
@pytest.mark.parametrize("exact3", [False, True], ids=["fast16", "exact"])
def test_auto_dispatch_shape_limits(B, N, T, d, exact3):
    """Shapes on both sides of the fused kernels' limits (N <= 208, T <= 28, d a multiple of 256:
    several 128-channel slices per wave at d = 768, 1024, 2048)."""
    from tests._hip import run_hip
    lens = sorted([T] + [max(1, T // 2)] * (B - 1), reverse=True)
    P = O.make_params(d, 9)
    V, Qs = O.make_inputs(B, N, T, d, 41, lens=lens, scale_q=(2.0 / d) ** 0.5)
    gv = torch.from_numpy(O.hash_normal((3, B, d), 5)).float()
    gq = torch.from_numpy(O.hash_normal((3, B, d), 6)).float()
    f = O.coattn_forward(V, Qs, P)
    g = O.coattn_backward(V, Qs, P, gv, gq)
    for layout in ("cm", "lm"):
        r = run_hip(V, Qs, P, gv, gq, impl="auto", layout=layout, exact3=exact3)
        assert (r["v"].cpu() - f["v"]).abs().max() < 1e-4 and (r["q"].cpu() - f["q"]).abs().max() < 1e-4
        for k in ("dV_phys", "dQ", "dW_v.weight", "dW_q.weight", "dw_v.weight", "dw_q.weight", "dW_v.bias", "dW_q.bias"):
            ref = g[k]
            assert (r[k].cpu() - ref).abs().max() <= 1e-4 * max(1e-3, ref.abs().max().item()), (layout, k)


def test_errors_are_loud():
    import vqa_amd
    from vqa_amd import _lib
    m = _mod(64)
    x = torch.zeros(2, 9, 64, device="cuda")
    with pytest.raises(RuntimeError):
        m(x, [torch.zeros(2, 5, 64, device="cuda"), torch.zeros(2, 6, 64, device="cuda"), torch.zeros(2, 5, 64, device="cuda")])
    with pytest.raises(RuntimeError):
        m(x.double(), [torch.zeros(2, 5, 64, device="cuda", dtype=torch.float64)] * 3)
    # C-ABI: fused kernels requested for an unsupported shape -> negative return, message set
    with pytest.raises(RuntimeError, match="fused"):
        vqa_amd.coattention(x, [torch.zeros(2, 5, 64, device="cuda")] * 3, m.W_v.weight, m.W_v.bias, m.W_q.weight,
                            m.W_q.bias, m.w_v.weight, m.w_v.bias, m.w_q.weight, m.w_q.bias, impl=_lib.IMPL_FUSED)


@pytest.mark.parametrize("exact3", [False, True], ids=["fast16", "exact"])
@pytest.mark.parametrize("L", [1, 2])
@pytest.mark.parametrize("impl", ["general", "auto"])
def test_fewer_than_three_levels(L, impl, exact3):
    """The module loops over whatever hierarchy it is given (model.py:372); L = 1, 2 also work."""
    from tests._hip import run_hip
    B, N, T, d = 3, 49, 26, 256
    P = O.make_params(d, 4)
    V, Qs = O.make_inputs(B, N, T, d, 51, lens=[26, 8, 1], scale_q=(2.0 / d) ** 0.5, L=L)
    gv = torch.from_numpy(O.hash_normal((L, B, d), 7)).float()
    gq = torch.from_numpy(O.hash_normal((L, B, d), 8)).float()
    if impl == "general" and not exact3:
        pytest.skip("the general-shape path has one arithmetic (exact): run once")
    f = O.coattn_forward(V, Qs, P)
    g = O.coattn_backward(V, Qs, P, gv, gq)
    for layout in ("cm", "lm"):                                  # (the exact mode's live-row paths with one and two levels too)
        r = run_hip(V, Qs, P, gv, gq, impl=impl, layout=layout, exact3=exact3)
        assert (r["v"].cpu() - f["v"]).abs().max() < 1e-4 and (r["q"].cpu() - f["q"]).abs().max() < 1e-4
        for k in ("dV_phys", "dQ", "dW_v.weight", "dW_q.weight", "dw_v.weight", "dw_q.weight", "dW_v.bias", "dW_q.bias"):
            assert (r[k].cpu() - g[k]).abs().max() <= 1e-4 * max(1e-3, g[k].abs().max().item()), (layout, k)


def test_trainer_validate_forward_only_path():
    """compute_validation_metrics (main.py:290-351): eval() + no_grad forward through the HIP op."""
    from vqa_amd import train as T
    torch.manual_seed(0)
    model = T.build_model("attention", 60, 10).cuda()
    tr = T.Trainer(model, 1e-4, torch.device("cuda:0"))
    batches = []
    for i in range(2):
        b = T.synthetic_batch(4, (64, 64), 26, 60, 11, seed=10 + i)
        im, qu, la, ln = T.sort_batch(b["image"], b["question"], b["label"], b["ques_len"])
        batches.append((im.cuda(), qu.cuda(), ln, la.cuda()))
    m = tr.validate(batches)
    assert 0.0 <= m["accuracy"] <= 100.0 and m["loss"] > 0 and model.training


@pytest.mark.parametrize("shape", [(2, 49, 26, 2048, "general"), (3, 196, 26, 512, "auto"), (160, 49, 26, 2048, "fused")],
                         ids=lambda s: "B%d_N%d_T%d_d%d_%s" % s)
def test_bf16_mfma_projections(shape):
    """BASELINE config 4 shape (7x7x2048 features: at B = 2 on the general-shape kernels, and at the FULL batch of 160
    through the fused kernels with the projections on gemm_bf.hip / gemm_tn's single-piece mode) and the cfg-2 shape with
    the projections on the bf16 MFMA (COATTN_FLAG_BF16_PROJ): bf16-rounded operands, one MFMA per product, fp32
    accumulation -> bf16 tolerance (stated: 3e-2 absolute on v / q, 5e-2 of max|.| on the gradients)."""
    from tests._hip import run_hip
    B, N, T, d, impl = shape
    P = O.make_params(d, 12)
    V, Qs = O.make_inputs(B, N, T, d, 61, lens=sorted([26] + [5] * (B - 1), reverse=True), scale_q=(2.0 / d) ** 0.5)
    gv = torch.from_numpy(O.hash_normal((3, B, d), 3)).float()
    gq = torch.from_numpy(O.hash_normal((3, B, d), 4)).float()
    r = run_hip(V, Qs, P, gv, gq, impl=impl, bf16_proj=True, layout="lm" if B > 8 else "cm")
    x = run_hip(V, Qs, P, gv, gq, impl=impl, bf16_proj=False, layout="lm" if B > 8 else "cm")
    sub = [0, B // 2, B - 1] if B > 8 else list(range(B))             # (the float64 oracle on a subset at full size)
    f = O.coattn_forward(V[sub], [q[sub] for q in Qs], P)
    r["v"], r["q"] = r["v"][:, sub], r["q"][:, sub]
    # the projections themselves: P_v within bf16 rounding of the exact product, and not identical to fp32
    rel = ((r["P_v"] - x["P_v"]).abs().max() / x["P_v"].abs().max()).item()
    assert 1e-5 < rel < 2e-2, rel
    assert (r["v"].cpu() - f["v"]).abs().max() < 3e-2 and (r["q"].cpu() - f["q"]).abs().max() < 3e-2
    assert all(torch.isfinite(t).all() for t in r.values())
    # backward in the same mode: the d x d gradient contractions run on the bf16 MFMA as well.  Checker: the ORACLE in
    # float64 (not the HIP fp32 run) -- dQ / dV on the sample subset (they are per sample), the parameter
    # gradients on the full batch.  Stated bounds of this mode (every operand of every product rounded to bf16 = 8
    # significant bits, fp32 accumulation): max error 8e-2 of max|.|, relative L2 4e-2 (observed <= 5e-2 / 2.7e-2).
    d64 = lambda t: t.double()                                    # noqa: E731
    P64 = {k: d64(v) for k, v in P.items()}
    gs = O.coattn_backward(d64(V[sub]), [d64(q[sub]) for q in Qs], P64, d64(gv[:, sub]), d64(gq[:, sub]))
    gf = gs if len(sub) == B else O.coattn_backward(d64(V), [d64(q) for q in Qs], P64, d64(gv), d64(gq))
    checks = {"dQ": (r["dQ"][:, sub], gs["dQ"]), "dV_phys": (r["dV_phys"][sub], gs["dV_phys"]),
              "dW_v.weight": (r["dW_v.weight"], gf["dW_v.weight"]), "dW_q.weight": (r["dW_q.weight"], gf["dW_q.weight"]),
              "dW_v.bias": (r["dW_v.bias"], gf["dW_v.bias"]), "dw_v.weight": (r["dw_v.weight"], gf["dw_v.weight"])}
    for k, (got, ref) in checks.items():
        got = got.cpu().double()
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        l2 = ((got - ref).norm() / ref.norm()).item()
        print("bf16 mode %s vs float64 oracle: max err / max|.| %.3e, relative L2 %.3e" % (k, err, l2))
        assert err < 8e-2 and l2 < 4e-2, (k, err, l2)
    assert (r["dW_q.weight"] - x["dW_q.weight"]).abs().max().item() > 0      # ... and it IS another arithmetic than fp32


@pytest.mark.parametrize("shape", [(160, 49, 26, 2048), (16, 196, 26, 1024)], ids=lambda s: "B%d_N%d_T%d_d%d" % s)
def test_bf16_mode_stores_its_gemm_only_gradients_as_bf16(shape):
    """Reduced-precision mode at config 4's full size (and at the 14 x 14 grid) with a frozen image encoder (no dV): bwd_nat32 stores dP_v / dP_q
    as bf16 and the three GEMMs that consume them read them as stored.  Every value is rounded at the same place as with
    fp32 storage, so dQ and dW_q are BIT-identical to the run that keeps them in fp32 (the one that also asks for dV);
    dW_v sums the three levels after the rounding instead of before it (bf16 tolerance)."""
    from tests._hip import run_hip
    B, N, T, d = shape
    P = O.make_params(d, 12)
    V, Qs = O.make_inputs(B, N, T, d, 62, lens=sorted([26] + [5] * (B - 1), reverse=True), scale_q=(2.0 / d) ** 0.5)
    gv = torch.from_numpy(O.hash_normal((3, B, d), 5)).float()
    gq = torch.from_numpy(O.hash_normal((3, B, d), 6)).float()
    a = run_hip(V, Qs, P, gv, gq, impl="fused", bf16_proj=True, layout="lm", need_dv=False)
    b = run_hip(V, Qs, P, gv, gq, impl="fused", bf16_proj=True, layout="lm", need_dv=True)
    for k in ("dQ", "dW_q.weight", "dW_q.bias", "dW_v.bias", "dw_v.weight", "dw_q.weight"):
        assert torch.equal(a[k], b[k]), k
    l2 = ((a["dW_v.weight"] - b["dW_v.weight"]).double().norm() / b["dW_v.weight"].double().norm()).item()
    print("dW_v, levels summed after / before the bf16 rounding: relative L2 %.3e" % l2)
    assert 0 < l2 < 1e-2
    a2 = run_hip(V, Qs, P, gv, gq, impl="fused", bf16_proj=True, layout="lm", need_dv=False)
    assert torch.equal(a["dW_v.weight"], a2["dW_v.weight"])           # repeatable bit for bit


def test_runs_on_the_callers_stream():
    """The C-ABI enqueues on the stream it is given (torch's current stream), not the default one."""
    B, N, T, d = 4, 49, 26, 512
    m = _mod(d)
    V, Qs = O.make_inputs(B, N, T, d, 71, lens=[26, 12, 3, 1], scale_q=(2.0 / d) ** 0.5)
    x = V.cuda().permute(0, 2, 1)
    Qg = [q.cuda().requires_grad_(True) for q in Qs]
    v0, q0 = m(x, Qg)
    sum(t.sum() for t in v0 + q0).backward()
    g0 = [t.grad.clone() for t in Qg]
    for t in Qg:
        t.grad = None
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        big = torch.randn(4096, 4096, device="cuda") @ torch.randn(4096, 4096, device="cuda")   # keep the stream busy
        v1, q1 = m(x, Qg)
        sum(t.sum() for t in v1 + q1).backward()
    side.synchronize()
    assert torch.isfinite(big).all()
    for l in range(3):
        assert torch.equal(v0[l], v1[l]) and torch.equal(q0[l], q1[l])
        assert torch.equal(g0[l], Qg[l].grad)


@pytest.mark.parametrize("fast", [True, False], ids=["fast16", "exact"])
@pytest.mark.parametrize("N", [49, 196])
def test_large_batch_is_sample_independent(N, fast):
    """4x the configured batch (B = 640): every sample's outputs and input gradients are the ones it
    gets in a batch of 8 (bitwise for the forward: the per-sample reduction order does not depend on
    the grid), parameter gradients are the sum over samples."""
    torch.manual_seed(1)
    B, T, d = 640, 26, 512
    m = _mod(d)
    m.fast_products = fast
    V = torch.randn(B, d, N, device="cuda").clamp_min_(0)
    lens = [T] + [1 + (5 * i) % T for i in range(B - 1)]
    mask = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).float().unsqueeze(-1).cuda()
    Qs = [(torch.randn(B, T, d, device="cuda") * (2.0 / d) ** 0.5 * mask) for _ in range(3)]
    gv = torch.randn(3, B, d, device="cuda"); gq = torch.randn(3, B, d, device="cuda")

    def run(sl):
        for p in m.parameters():
            p.grad = None
        q = [t[sl].clone().requires_grad_(True) for t in Qs]
        x = V[sl].permute(0, 2, 1).clone().requires_grad_(True)
        v_o, q_o = m(x, q)
        loss = sum((v_o[l] * gv[l, sl]).sum() + (q_o[l] * gq[l, sl]).sum() for l in range(3))
        loss.backward()
        return (torch.stack(v_o).detach(), torch.stack(q_o).detach(), x.grad, torch.stack([t.grad for t in q]),
                m.W_v.weight.grad.clone(), m.W_q.weight.grad.clone())

    big = run(slice(0, B))
    for s0 in (0, 320, 632):
        small = run(slice(s0, s0 + 8))
        assert torch.equal(big[0][:, s0:s0 + 8], small[0]) and torch.equal(big[1][:, s0:s0 + 8], small[1])
        assert (big[2][s0:s0 + 8] - small[2]).abs().max().item() < 1e-5 * max(1.0, small[2].abs().max().item())
        assert (big[3][:, s0:s0 + 8] - small[3]).abs().max().item() < 1e-5 * max(1.0, small[3].abs().max().item())
    # parameter gradients: B = 640 equals the sum of its two halves
    h1 = run(slice(0, 320)); h2 = run(slice(320, 640))
    for k in (4, 5):
        ref = h1[k] + h2[k]
        assert (big[k] - ref).abs().max().item() < 1e-4 * ref.abs().max().item()


def test_large_batch_backward_workspace():
    """B large enough that the da_v partials ([B][d/64][3][N]) outgrow the 32 d^2 split-K scratch they share
    (B * 3 * N > 2048 d): the workspace plan must cover them (round-1 advisor finding).  Fused vs general-shape
    kernels on the same inputs; the workspace is followed by a canary region that must stay untouched."""
    import ctypes as C
    from tests._hip import run_hip
    from vqa_amd import _lib
    B, N, T, d = 1024, 196, 26, 256
    assert B * 3 * N > 2048 * d
    lens = sorted([T] + [3 + (5 * i) % (T - 2) for i in range(B - 1)], reverse=True)
    P = O.make_params(d, 13)
    V, Qs = O.make_inputs(B, N, T, d, 57, lens=lens, scale_q=(2.0 / d) ** 0.5)
    gv = torch.from_numpy(O.hash_normal((3, B, d), 15)).float()
    gq = torch.from_numpy(O.hash_normal((3, B, d), 16)).float()
    a = run_hip(V, Qs, P, gv, gq, impl="fused", need_dv=False)
    b = run_hip(V, Qs, P, gv, gq, impl="general", need_dv=False)
    for k in ("dQ", "dW_v.weight", "dW_q.weight", "dw_v.weight", "dw_q.weight", "dW_v.bias", "dW_q.bias"):
        assert (a[k] - b[k]).abs().max() <= 1e-4 * max(1e-3, b[k].abs().max().item()), k
    # the plan itself: bytes reported >= what bwd_dav_kernel writes behind the other regions
    _, _, bb = _lib.workspace_bytes(B, N, T, d, 3, _lib.IMPL_FUSED)
    assert bb // 4 >= B * (d // 64) * 3 * N


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(23, 196, 26, 512, "lm"), (9, 49, 26, 512, "cm"), (3, 100, 17, 1024, "lm"),
                                   (640, 196, 26, 512, "cm"), (640, 196, 26, 512, "lm"), (320, 49, 26, 512, "cm")])
@pytest.mark.parametrize("fast", [True, False], ids=["fast16", "exact"])
def test_repeated_runs_are_bitwise_identical(shape, fast):
    """no float atomics, fixed summation orders -- forward outputs and every gradient (image features
    included) of repeated runs on the same inputs are bit-for-bit the same (also a check on the LDS-only barriers and
    counted waits of the fused kernels: a race would show here).  The large batches are there because a race needs a busy
    chip: the missing order between a ring slot's read-back and its refill in the channel-major phase 1 (round 4) showed in
    one launch out of eight at B = 640 and in one out of seventy at B = 160."""
    import vqa_amd
    B, N, T, d, lay = shape
    torch.manual_seed(7)
    co = vqa_amd.ParallelCoAttention(d).cuda()
    co.fast_products = fast
    x = (torch.randn(B, d, N, device="cuda") * 0.5).permute(0, 2, 1)
    if lay == "lm":
        x = x.contiguous()
    x.requires_grad_(True)
    Qs = [(torch.randn(B, T, d, device="cuda") * 0.5).requires_grad_(True) for _ in range(3)]
    ref = None
    for _ in range(12 if B < 100 else 30):
        for t in [x] + Qs + list(co.parameters()):
            t.grad = None
        v, q = co(x, Qs)
        (sum((a * a).sum() for a in v) + sum((a * (a + 1)).sum() for a in q)).backward()
        outs = [t.detach().clone() for t in v + q] + [x.grad.clone()] + [t.grad.clone() for t in Qs] + \
               [p.grad.clone() for p in co.parameters() if p.grad is not None]
        if ref is None:
            ref = outs
        else:
            assert all(torch.equal(a, b) for a, b in zip(ref, outs))


@pytest.mark.gpu
@pytest.mark.parametrize("layout,N", [("lm", 196), ("cm", 196), ("lm", 49)])
@pytest.mark.parametrize("exact3", [False, True], ids=["fast16", "exact"])
def test_gradients_do_not_depend_on_whether_dV_is_requested(layout, N, exact3):
    """Without dV (the frozen-encoder default) the weight-gradient kernel adds the three levels of dP_v while it
    stages them; with dV a separate pass sums them first.  Same additions in the same order: every other gradient
    must be bit-for-bit the same either way, on both feature layouts."""
    from tests._hip import run_hip
    B, T, d = 24, 26, 512
    P = O.make_params(d, 21)
    V, Qs = O.make_inputs(B, N, T, d, 31, lens=sorted([T] + [2 + (3 * i) % 24 for i in range(B - 1)], reverse=True),
                          scale_q=(2.0 / d) ** 0.5)
    gv = torch.from_numpy(O.hash_normal((3, B, d), 41)).float()
    gq = torch.from_numpy(O.hash_normal((3, B, d), 42)).float()
    a = run_hip(V, Qs, P, gv, gq, impl="fused", layout=layout, need_dv=True, exact3=exact3)
    b = run_hip(V, Qs, P, gv, gq, impl="fused", layout=layout, need_dv=False, exact3=exact3)
    for k in ("dQ", "dW_v.weight", "dW_q.weight", "dw_v.weight", "dw_q.weight", "dW_v.bias", "dW_q.bias"):
        assert torch.equal(a[k], b[k]), k


@pytest.mark.gpu
def test_question_features_not_16_byte_aligned():
    """Q_l at a 4-byte offset: the pre-split-weight GEMMs (which need 16-byte rows) must step aside in the forward AND in
    the backward's dQ projection -- whose weight image only exists when the forward took that kernel."""
    import vqa_amd
    B, N, T, d = 6, 196, 26, 512
    torch.manual_seed(3)
    co = vqa_amd.ParallelCoAttention(d).cuda()
    x = (torch.randn(B, N, d, device="cuda") * 0.5)
    base = [torch.randn(B * T * d + 1, device="cuda") * 0.5 for _ in range(3)]
    gv = torch.randn(3, B, d, device="cuda"); gq = torch.randn(3, B, d, device="cuda")

    def run(qs):
        for p in co.parameters():
            p.grad = None
        qs = [q.detach().requires_grad_(True) for q in qs]
        v, q = co(x, qs)
        (sum((v[l] * gv[l]).sum() + (q[l] * gq[l]).sum() for l in range(3))).backward()
        return torch.stack(v), torch.stack(q), torch.stack([t.grad for t in qs]), co.W_q.weight.grad.clone(), co.W_v.weight.grad.clone()

    off = [b[1:].view(B, T, d) for b in base]                    # data_ptr % 16 == 4
    assert all(t.data_ptr() % 16 == 4 for t in off)
    a = run(off)
    b = run([t.clone() for t in off])                            # the same values, aligned
    for u, w in zip(a, b):
        assert torch.isfinite(u).all()
        assert (u - w).abs().max().item() <= 2e-5 * max(1e-3, w.abs().max().item())


def test_profile_marks_of_forward_and_backward():
    """coattn_profile_begin / _end (bench.py's per-kernel legs): one mark per launch group of the calls in between, positive
    times that add up to about the span of the calls; without a begin the end call is an error, not a crash."""
    import ctypes as C
    import torch
    from oracle import coattn_oracle as O
    from vqa_amd import _lib
    lib = _lib.load()
    us = (C.c_float * 48)()
    names = C.create_string_buffer(1024)
    assert lib.coattn_profile_end(us, names, 1024, 48) == -1 and b"coattn_profile_begin" in lib.coattn_last_error()
    B, N, T, d, L = 8, 49, 26, 512, 3
    P = O.make_params(d, 5)
    V, Qs = O.make_inputs(B, N, T, d, 5, lens=[T] * B, scale_q=(2.0 / d) ** 0.5)
    dev = torch.device("cuda:0")
    Vb = V.permute(0, 2, 1).contiguous().to(dev)
    Qd = [q.to(dev) for q in Qs]
    keys = ("W_v.weight", "W_v.bias", "W_q.weight", "W_q.bias", "w_v.weight", "w_v.bias", "w_q.weight", "w_q.bias")
    ps = [P[k].to(dev).contiguous() for k in keys]
    sb, fb, bb = _lib.workspace_bytes(B, N, T, d, L)
    saved = torch.empty(sb // 4, device=dev); ws = torch.empty(max(fb, bb) // 4, device=dev)
    v = torch.empty(L, B, d, device=dev); q = torch.empty(L, B, d, device=dev)
    g = torch.ones(L, B, d, device=dev)
    dQ = [torch.empty_like(t) for t in Qd]; grads = [torch.empty_like(t) for t in ps]
    qptr = (C.c_void_p * L)(*[t.data_ptr() for t in Qd]); dqptr = (C.c_void_p * L)(*[t.data_ptr() for t in dQ])
    p = _lib.Params(*[t.data_ptr() for t in ps]); pg = _lib.ParamGrads(*[t.data_ptr() for t in grads])
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    vs = (N * d, d, 1)
    for _ in range(2):                                       # (the first call pays the one-time kernel attributes)
        _lib.check(lib.coattn_profile_begin(st), "coattn_profile_begin")
        _lib.check(lib.coattn_forward(Vb.data_ptr(), *vs, qptr, C.byref(p), v.data_ptr(), q.data_ptr(), saved.data_ptr(),
                                      ws.data_ptr(), B, N, T, d, L, _lib.F32, 0, st), "coattn_forward")
        _lib.check(lib.coattn_backward(Vb.data_ptr(), *vs, qptr, C.byref(p), saved.data_ptr(), g.data_ptr(), g.data_ptr(), None,
                                       0, 0, 0, dqptr, C.byref(pg), 0, ws.data_ptr(), B, N, T, d, L, _lib.F32, 0, st),
                   "coattn_backward")
        n = lib.coattn_profile_end(us, names, 1024, 48)
    marks = names.value.decode().split("\n")
    assert n == len(marks) and n >= 7, (n, marks)
    # (the partial sums of the weight gradients ride in the dQ kernel's launch: bwd_dq is the last mark)
    assert marks[:3] == ["wsplit", "projections", "coattn_fwd32"] and "bwd_nat32" in marks and marks[-1] == "bwd_dq"
    assert all(us[i] > 0 for i in range(n)) and sum(us[i] for i in range(n)) < 5e4
    assert torch.isfinite(v).all() and all(torch.isfinite(t).all() for t in grads)


def test_large_feature_magnitudes_stay_finite():
    """Tolerance mode (COATTN_FLAG_FAST16: forward-side contractions on two FP16 pieces, include/coattn.h): features far
    beyond fp16's range must not turn into inf - inf = NaN -- the conversions saturate -- AND the event is reported
    (coattn_status = -4); in the exact mode (flags = 0) the values are carried exactly and nothing is reported."""
    import vqa_amd
    from tests import _hip
    from tests._hip import run_hip
    from tests import _golden as G
    torch.manual_seed(77)
    V, Qs, P, gv, gq = G.build_case("g3_n49_ragged", torch.float32)
    V = V.clone()
    V[:, :, 0] *= 1.0e5                                   # one location of every sample, all channels: up to ~1e6
    V[0, 3, :] = 1.0e30
    for exact in (False, True):
        r = run_hip(V, Qs, P, gv, gq, impl="fused", layout="lm", exact3=exact)
        for k, t in r.items():
            assert torch.isfinite(t).all(), (exact, k)
        rc, a_act, a_w = _hip.last_status
        assert rc == (0 if exact else -4), (exact, _hip.last_status)
        if not exact:
            assert a_act >= 1.0e30 and b"65504" in vqa_amd._lib.load().coattn_last_error()


@pytest.mark.parametrize("layout", ["lm", "cm"])
def test_fp16_piece_range_report_and_exact_mode_value(layout):
    """VERDICT r4 / ADVICE r4: outside the FP16-piece range the tolerance mode clamps -- that must not be silent.  One image
    feature of magnitude 2e5 (and, separately, one projection weight of 300): the tolerance mode's status word is set and
    names the magnitude; the exact mode -- flags = 0, the C-ABI's default -- reports nothing and computes the REFERENCE's
    value (float64 oracle on the same inputs, 2e-5 of max|.|: fp32 rounding of values spanning 2e5).  A non-finite feature: reported in the tolerance mode (where
    the contractions clamp it), and non-finite outputs in the exact mode, as the reference's fp32 bmm gives."""
    from tests import _hip
    from tests._hip import run_hip
    from tests import _golden as G
    N = 196 if layout == "cm" else 49
    B, T, d = 4, 26, 512
    P = O.make_params(d, 31)
    V, Qs = O.make_inputs(B, N, T, d, 33, lens=[26, 17, 4, 1], scale_q=(2.0 / d) ** 0.5)
    gv = torch.from_numpy(O.hash_normal((3, B, d), 7)).float()
    gq = torch.from_numpy(O.hash_normal((3, B, d), 8)).float()
    r = run_hip(V, Qs, P, gv, gq, impl="fused", layout=layout)
    assert _hip.last_status[0] == 0 and _hip.last_status[1] == 0.0      # ordinary magnitudes: nothing to report
    assert 0 < _hip.last_status[2] < 65504.0                           # (largest |256 W| ~ 256 / sqrt(d))
    Vb = V.clone()
    Vb[1, 7, 5] = 2.0e5
    d64 = lambda t: t.double()                                          # noqa: E731
    P64 = {k: d64(v) for k, v in P.items()}
    f = O.coattn_forward(d64(Vb), [d64(q) for q in Qs], P64)
    g = O.coattn_backward(d64(Vb), [d64(q) for q in Qs], P64, d64(gv), d64(gq))
    fast = run_hip(Vb, Qs, P, gv, gq, impl="fused", layout=layout)
    assert _hip.last_status[0] == -4 and abs(_hip.last_status[1] - 2.0e5) < 1.0, _hip.last_status
    exact = run_hip(Vb, Qs, P, gv, gq, impl="fused", layout=layout, exact3=True)
    assert _hip.last_status[0] == 0, _hip.last_status
    rel = lambda a, b: float((a.cpu().double() - b).abs().max() / b.abs().max())   # noqa: E731
    errs = {k: rel(exact[k], f[k]) for k in ("v", "q", "a_v", "a_q")}
    errs.update({k: rel(exact[k], g[k]) for k in ("dQ", "dV_phys", "dW_v.weight", "dW_q.weight", "dw_v.weight")})
    print("exact mode with a 2e5 feature (%s): errors vs float64 oracle" % layout, {k: "%.1e" % e for k, e in errs.items()})
    assert max(errs.values()) < 2e-5, errs
    # ... and the clamped run is NOT the reference's value: this is what the status word is for
    assert rel(fast["v"], f["v"]) > 1e-3 or rel(fast["dV_phys"], g["dV_phys"]) > 1e-3
    # a projection weight beyond the scaled weight image's range
    Pw = {k: v.clone() for k, v in P.items()}
    Pw["W_q.weight"][3, 9] = 300.0
    run_hip(V, Qs, Pw, gv, gq, impl="fused", layout=layout)
    assert _hip.last_status[0] == -4 and abs(_hip.last_status[2] - 300.0 * 256.0) < 1.0, _hip.last_status
    ew = run_hip(V, Qs, Pw, gv, gq, impl="fused", layout=layout, exact3=True)
    fw = O.coattn_forward(d64(V), [d64(q) for q in Qs], {k: d64(v) for k, v in Pw.items()})
    assert _hip.last_status[0] == 0 and rel(ew["v"], fw["v"]) < 1e-5 and rel(ew["q"], fw["q"]) < 1e-5
    # a non-finite feature
    Vi = V.clone()
    Vi[2, 11, 3] = float("inf")
    ri = run_hip(Vi, Qs, P, gv, gq, impl="fused", layout=layout)
    assert _hip.last_status[0] == -4 and _hip.last_status[1] == float("inf")      # (clamped inside the contractions: reported)
    assert torch.isfinite(ri["q"]).all()                                # (the attention maps stay finite ...)
    re_ = run_hip(Vi, Qs, P, gv, gq, impl="fused", layout=layout, exact3=True)
    assert not torch.isfinite(re_["v"][:, 2]).all()                     # the exact mode surfaces it, as the reference does


def test_range_report_is_sticky_across_forwards():
    """a clamped activation must stay visible until somebody LOOKS, not until the next forward rewrites the status
    words.  Tolerance mode: one forward with a 2e5 feature between ordinary ones (training forward with `saved`, an inference
    forward without, a second module in between) -> the first check_range() afterwards raises and names the magnitude, the
    next one is clean.  The exact mode (the default) reports nothing for the same inputs."""
    import vqa_amd
    B, N, T, d = 4, 49, 26, 512
    m, other = _mod(d), _mod(d, seed=4)
    V, Qs = O.make_inputs(B, N, T, d, 33, lens=[26, 17, 4, 1], scale_q=(2.0 / d) ** 0.5)
    Qg = [q.cuda() for q in Qs]
    good = V.cuda().permute(0, 2, 1).contiguous()
    bad = good.clone()
    bad[1, 7, 5] = 2.0e5
    vqa_amd.check_range()                                     # (whatever earlier tests left behind)
    for mod in (m, other):
        mod.fast_products = True
    m(good.clone().requires_grad_(True), Qg)
    m(bad.clone().requires_grad_(True), Qg)                   # the step that clamps
    for _ in range(3):
        m(good.clone().requires_grad_(True), Qg)              # later steps rewrite the status words of their own `saved`
    with torch.no_grad():
        m(good, Qg)                                           # validation pass (status words in the scratch workspace)
    other(good.clone().requires_grad_(True), Qg)              # another model
    with pytest.raises(vqa_amd.RangeError, match="2e\\+05|200000"):
        vqa_amd.check_range()
    vqa_amd.check_range()                                     # consumed: clean again
    m.fast_products = False
    m(bad.clone().requires_grad_(True), Qg)
    vqa_amd.check_range()


def test_trainer_defaults_to_the_references_arithmetic():
    """train.Trainer / train.py run fp32-accurate products unless asked otherwise (--opt_lvl 0 is the reference's
    fp32); the tolerance mode is the opt-in, and a range event there switches the trainer to exact."""
    from vqa_amd import train as T
    torch.manual_seed(0)
    model = T.build_model("attention", 60, 10).cuda()
    tr = T.Trainer(model, 1e-4, torch.device("cuda:0"))
    assert tr.precision == "exact" and model.co_attention.fast_products is False
    tr.set_precision("fast")
    assert model.co_attention.fast_products is True and tr.check_range() is True
    b = T.synthetic_batch(4, (224, 224), 26, 60, 11, seed=3)      # 7 x 7 grid: a shape the fused kernels take
    im, qu, la, ln = T.sort_batch(b["image"], b["question"], b["label"], b["ques_len"])
    with torch.no_grad():
        model.co_attention.W_q.weight[3, 9] = 300.0           # beyond the scaled FP16 weight image's range
    tr.step(im.cuda(), qu.cuda(), ln, la.cuda())
    tr.step(im.cuda(), qu.cuda(), ln, la.cuda())
    with pytest.warns(UserWarning, match="exact mode"):
        assert tr.check_range() is False
    assert tr.precision == "exact" and model.co_attention.fast_products is False


@pytest.mark.parametrize("pattern", ["sorted_pad_tails", "scattered", "none_zero", "one_level_all_zero", "negative_zero_and_nan"])
@pytest.mark.parametrize("layout", ["lm", "cm"])
def test_zero_question_rows_take_the_bias_path_bit_for_bit(pattern, layout):
    """Exact mode: question rows of exact zeros (the pad tokens, model.py:263 / :292-296) are flagged by the weight-split launch,
    their P_q rows written as (0 + b_q) * scale there, and the projection GEMM runs over the other rows only.  P_q must be
    BIT-IDENTICAL to the dense product of every row (coattn_linear_forward on the same rows, which has no row bitmap), for
    pad tails, zero rows scattered anywhere, no zero row at all, a whole level of zeros, rows of -0.0 (zero: skipped) and a
    row holding a NaN (not zero: computed); outputs and gradients equal the float64 oracle's as ever."""
    import ctypes as C
    from tests._hip import run_hip
    from vqa_amd import _lib
    B, N, T, d = 24, (49 if layout == "lm" else 52), 26, 512
    P = O.make_params(d, 17)
    lens = sorted([T] + [1 + (5 * i) % T for i in range(B - 1)], reverse=True)
    V, Qs = O.make_inputs(B, N, T, d, 91, lens=[T] * B, scale_q=(2.0 / d) ** 0.5)      # dense rows first
    Qs = [q.clone() for q in Qs]
    g = torch.Generator().manual_seed(5)
    if pattern == "sorted_pad_tails":
        for b, n in enumerate(lens):
            for q in Qs:
                q[b, n:] = 0.0
    elif pattern == "scattered":
        for q in Qs:
            q[torch.rand(B, T, generator=g) < 0.4] = 0.0
    elif pattern == "one_level_all_zero":
        Qs[1].zero_()
    elif pattern == "negative_zero_and_nan":
        Qs[0][3, 5] = -0.0
        Qs[2][7, 0:4] = -0.0
        Qs[1][2, 9, 17] = float("nan")
    gv = torch.from_numpy(O.hash_normal((3, B, d), 7)).float()
    gq = torch.from_numpy(O.hash_normal((3, B, d), 8)).float()
    r = run_hip(V, Qs, P, gv, gq, impl="fused", layout=layout, exact3=True)
    # the dense product of the same rows through the linear entry point (fused path: P_q is stored times 2 log2(e))
    lib = _lib.load()
    kPScale = 2.8853900817779268
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    wimg = torch.empty(lib.coattn_linear_workspace_bytes(d, d) // 4, device="cuda")
    Wq, bq = P["W_q.weight"].cuda().contiguous(), P["W_q.bias"].cuda().contiguous()
    for l in range(3):
        x = Qs[l].cuda().reshape(B * T, d).contiguous()
        y = torch.full((B * T, d), float("nan"), device="cuda")
        _lib.check(lib.coattn_linear_forward(x.data_ptr(), d, Wq.data_ptr(), bq.data_ptr(), y.data_ptr(), wimg.data_ptr(),
                                             B * T, d, d, kPScale, 0, st), "coattn_linear_forward")
        torch.cuda.synchronize()
        got = r["P_q"][l].reshape(B * T, d)
        same = (got == y) | (torch.isnan(got) & torch.isnan(y))
        assert bool(same.all()), (pattern, l, int((~same).sum()))
    if pattern == "negative_zero_and_nan":
        return                                                   # (a NaN feature: nothing to compare with the oracle)
    d64 = lambda t: t.double()                                   # noqa: E731
    P64 = {k: d64(v) for k, v in P.items()}
    f = O.coattn_forward(d64(V), [d64(q) for q in Qs], P64)
    b = O.coattn_backward(d64(V), [d64(q) for q in Qs], P64, d64(gv), d64(gq))
    for k in ("v", "q", "a_v", "a_q"):
        assert (r[k].cpu().double() - f[k]).abs().max().item() < 2e-5, (pattern, k)
    for k in ("dQ", "dV_phys", "dW_q.weight", "dW_v.weight", "dW_q.bias"):
        ref = b[k]
        assert ((r[k].cpu().double().reshape(ref.shape) - ref).abs().max() / ref.abs().max()).item() < 2e-5, (pattern, k)


@pytest.mark.parametrize("B", [315, 330, 520])
def test_live_row_paths_on_both_sides_of_their_limits(B):
    """The exact mode's live-row machinery has two size limits: the device-planned weight gradients gather through an LDS map of at
    most 8,192 question rows per level (B T <= 8192: B = 315 is the last batch inside, with the largest map), the row bitmap
    itself has at most 512 words per level (B T <= 16,384).  B = 330 and 520 run the compacted P_q projection with the host's
    dense split-K plan behind it.  Fused against the general-shape kernels (which know nothing of pad rows) on the same inputs."""
    from tests._hip import run_hip
    N, T, d = 49, 26, 512
    lens = sorted([T] + [1 + (7 * i) % T for i in range(B - 1)], reverse=True)
    P = O.make_params(d, 23)
    V, Qs = O.make_inputs(B, N, T, d, 67, lens=lens, scale_q=(2.0 / d) ** 0.5)
    gv = torch.from_numpy(O.hash_normal((3, B, d), 25)).float()
    gq = torch.from_numpy(O.hash_normal((3, B, d), 26)).float()
    a = run_hip(V, Qs, P, gv, gq, impl="fused", layout="lm", need_dv=False, exact3=True)
    b = run_hip(V, Qs, P, gv, gq, impl="general", layout="lm", need_dv=False, exact3=True)
    assert (a["v"] - b["v"]).abs().max().item() < 1e-5 and (a["q"] - b["q"]).abs().max().item() < 1e-5
    for k in ("dQ", "dW_v.weight", "dW_q.weight", "dw_v.weight", "dw_q.weight", "dW_v.bias", "dW_q.bias"):
        assert (a[k] - b[k]).abs().max().item() <= 2e-5 * max(1e-3, b[k].abs().max().item()), (B, k)
    a2 = run_hip(V, Qs, P, gv, gq, impl="fused", layout="lm", need_dv=False, exact3=True)
    for k in ("v", "q", "dQ", "dW_v.weight", "dW_q.weight"):
        assert torch.equal(a[k], a2[k]), (B, k)                          # repeatable bit for bit
