"""GPU: seeded sweep of shapes through the C-ABI against the CPU oracle (float64), both kernel
families.  Shapes cover what the dispatch has to get right: sizes that are not multiples of any tile
(N, T, d odd or tiny), one or two levels, single samples, lengths down to 1, and shapes just inside /
outside the fused kernels' limits (coattn_fused_supported).  Tolerances as tests/test_gpu_parity.py."""
import random

import pytest
import torch

from oracle import coattn_oracle as O

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-4
GRAD_TOL = 1e-4


def _shapes():
    rng = random.Random(20261003)
    out = [(1, 1, 1, 4, 1), (2, 3, 2, 12, 2), (1, 208, 28, 256, 3), (2, 209, 26, 256, 3), (2, 196, 29, 256, 3),
           (3, 49, 26, 768, 3), (1, 196, 26, 1024, 2),
           # every tile count of the fused kernels' unit loops (1 .. 7 tiles of 32 locations; both tile-count templates)
           (2, 32, 5, 512, 3), (2, 33, 26, 512, 1), (1, 65, 3, 512, 2), (2, 97, 26, 512, 3), (1, 129, 7, 256, 3),
           (2, 161, 26, 512, 2), (1, 193, 28, 512, 3),
           # the widest hidden sizes the fused kernels take (4 and 8 channel slices per wave)
           (1, 196, 26, 2048, 3), (1, 64, 28, 4096, 1),
           # small grids on location-major features: the forward kernel attends the image features itself (two channel sweeps)
           (2, 50, 26, 1024, 3), (3, 37, 9, 512, 2)]
    for _ in range(14):
        d = rng.choice([4, 20, 36, 64, 100, 256, 512])
        out.append((rng.randint(1, 5), rng.randint(1, 210), rng.randint(1, 30), d, rng.randint(1, 3)))
    return out


# exact mode (flags = 0: the reference's arithmetic, fp32-accurate products) is held tighter than the contract
EXACT_TOL = 2e-5
# dw_v.bias / dw_q.bias are sums of softmax-Jacobian rows -- exactly 0 in real arithmetic -- so what any fp32 implementation returns
# is its summation noise; they are compared on this absolute scale (tools/fuzz_wide.py raises it for batches of hundreds)
BIAS_SCALE = 1.0


@pytest.mark.parametrize("exact3", [False, True], ids=["fast16", "exact"])
@pytest.mark.parametrize("shape", _shapes(), ids=lambda s: "B%d_N%d_T%d_d%d_L%d" % s)
def test_random_shape_vs_oracle(shape, exact3):
    import vqa_amd
    from tests._hip import run_hip
    B, N, T, d, L = shape
    seed = 1000 + B * 7 + N * 13 + T * 17 + d
    rng = random.Random(seed)
    lens = sorted([T] + [rng.randint(1, T) for _ in range(B - 1)], reverse=True)
    P = O.make_params(d, seed)
    V, Qs = O.make_inputs(B, N, T, d, seed, lens=lens, scale_q=(2.0 / d) ** 0.5, L=L)
    gv = torch.from_numpy(O.hash_normal((L, B, d), seed + 7)).float()
    gq = torch.from_numpy(O.hash_normal((L, B, d), seed + 8)).float()
    P64 = {k: v.double() for k, v in P.items()}
    f = O.coattn_forward(V.double(), [q.double() for q in Qs], P64)
    b = O.coattn_backward(V.double(), [q.double() for q in Qs], P64, gv.double(), gq.double())
    # (the general-shape kernels have one arithmetic -- exact -- and are swept under the exact id only)
    impls = ["general"] if exact3 else []
    if vqa_amd._lib.load().coattn_fused_supported(B, N, T, d, L, 0):
        impls.append("fused")
    if not impls:
        pytest.skip("general-shape path only: one arithmetic (exact), swept under the exact id")
    fwd_tol, grad_tol = (EXACT_TOL, EXACT_TOL) if exact3 else (FWD_TOL, GRAD_TOL)
    worst = 0.0
    from tests._hip import LAYOUTS
    for impl, layout in [(i, lay) for i in impls for lay in LAYOUTS]:
        r = run_hip(V, Qs, P, gv, gq, impl=impl, layout=layout, exact3=exact3)
        impl = impl + "/" + layout
        for k in ("v", "q", "C", "a_v", "a_q"):
            err = (r[k].double().cpu() - f[k]).abs().max().item()
            assert err < fwd_tol, (impl, k, err)
            worst = max(worst, err)
        grads = {"dV_phys": b["dV_phys"], "dQ": b["dQ"]}
        grads.update({"d" + k: b["d" + k] for k in O.PARAM_KEYS})
        for k, ref in grads.items():
            got = r[k].double().cpu().reshape(ref.shape)
            assert torch.isfinite(got).all(), (impl, k)
            scale = max(ref.abs().max().item(), BIAS_SCALE if k in ("dw_v.bias", "dw_q.bias") else 1e-30)
            err = (got - ref).abs().max().item() / scale
            assert err < grad_tol, (impl, k, err)
            worst = max(worst, err)
    print("shape", shape, "exact" if exact3 else "fast16", impls, "worst error %.1e" % worst)
