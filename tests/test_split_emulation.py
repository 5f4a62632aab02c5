"""Error budget of the mixed-width bf16 split ("widths"), emulated on the CPU.

The fp32 mode of the HIP kernels multiplies fp32 operands on the bf16 MFMA by splitting each value into bf16
pieces (hi + mid + lo, exact) and summing partial products in fp32.  The affinity and the projections
(model.py:377, :380-384) keep all three pieces (six products); the H_v / H_q contractions against C and the
gradient contractions of the backward run on TWO pieces (hi + mid; products hi*mid, mid*hi, hi*hi).  This test restates the path in float64 with exactly
those operand truncations and dropped partial products and holds it to the reference's goldens at the
contract's 1e-4 -- so that the widths are pinned by a test that runs without a GPU, and a change of the table
(WIDTHS) shows its cost here first.  `python -m tests.test_split_emulation` prints the cost of each row."""
import numpy as np
import pytest
import torch

from . import _golden as G

# contraction -> pieces per operand.  3: six partial products of bf16 pieces (error ~2^-24 per product), 2: three partial
# products (operands truncated to 16 significand bits, mid*mid dropped: ~2^-16 relative per product), "2h": two FP16
# pieces (22 significand bits, three partial products, ~2^-22; fp16's range), "2hs": the same with the weight operand
# scaled by 256 (fused.h kF16WScale: the lo pieces of ~0.04-sized weights would be subnormal otherwise)
WIDTHS = {
    "affinity": "2h",   # A = Q V^T                                   coattn_fwd32 phase 1
    "proj": "2hs",      # P_v = V W_v^T                               gemm_w   (two BF16 pieces: H_q off by 2e-4, see below)
    "proj_q": "2hs",    # P_q = Q W_q^T                               gemm_w
    "h": "2h",          # C^T P_q, C P_v                              coattn_fwd32 phase 2
    "bwd": 2,           # recomputed C^T P_q, C dZ_v, C^T dZ_q, dC    bwd_nat32 / bwd_dc32
    "dq": 2,            # dA V                                        bwd_dq32(x)
    "gemm_bwd": 2,      # dP_q W_q, dW_v, dW_q                        gemm_tn launch
    "dv": 3,            # dV terms (general GEMM, exact split)
}
EXACT = {k: 3 for k in WIDTHS}


def _pieces(x64, n, dt=torch.bfloat16):
    """fp32 value -> its first n bf16 (fp16) pieces (round to nearest even at every step), as float64 tensors."""
    r = x64.to(torch.float32)
    out = []
    for _ in range(n):
        p = r.to(dt).to(torch.float32)
        out.append(p.double())
        r = r - p                                   # exact in fp32
    return out


def mm(a, b, n):
    """a @ b as the n-piece kernels compute it (fp32 accumulation not modelled: float64 sums)."""
    if n == "2x3":                                  # a on two pieces, b (the weight) on three: ah bh + am bh + ah bm + ah bl
        pa, pb = _pieces(a, 2), _pieces(b, 3)
        return (pa[0] + pa[1]) @ pb[0] + pa[0] @ (pb[1] + pb[2])
    if n in ("2h", "2hs"):                          # two fp16 pieces (hi + lo: 22 significand bits, fp16's range)
        sc = 256.0 if n == "2hs" else 1.0           # ("2hs": b is the weight, held as 256 W)
        pa, pb = _pieces(a, 2, torch.float16), _pieces(b * sc, 2, torch.float16)
        return ((pa[0] + pa[1]) @ pb[0] + pa[0] @ pb[1]) / sc
    pa, pb = _pieces(a, n), _pieces(b, n)
    if n == 1:
        return pa[0] @ pb[0]
    if n == 2:
        return (pa[0] + pa[1]) @ pb[0] + pa[0] @ pb[1]
    sa, sb = pa[0] + pa[1] + pa[2], pb[0] + pb[1] + pb[2]
    return sa @ sb - (pa[1] @ pb[2] + pa[2] @ pb[1] + pa[2] @ pb[2])


def f32(x):
    return x.to(torch.float32).double()


def emulate(V_phys, Qs, P, gv, gq, W):
    """Forward + hand-derived backward (oracle/coattn_oracle.py) with every stored intermediate rounded to
    fp32 and every contraction at its width."""
    Wv, bv, Wq, bq = (P[k].double() for k in ("W_v.weight", "W_v.bias", "W_q.weight", "W_q.bias"))
    wv, cv, wq, cq = P["w_v.weight"][0].double(), P["w_v.bias"].double(), P["w_q.weight"][0].double(), P["w_q.bias"].double()
    V_phys = V_phys.double()
    Vn = V_phys.permute(0, 2, 1)
    Pv = f32(mm(Vn, Wv.T, W["proj"]) + bv)
    res = {k: [] for k in ("v", "q", "C", "a_v", "a_q", "H_q")}
    g = {k: torch.zeros_like(P[k].double()) for k in G.O.PARAM_KEYS}
    dVn = torch.zeros_like(Vn)
    dPv_tot = torch.zeros_like(Pv)
    dQs = []
    for l, Q in enumerate(Qs):
        Q = Q.double()
        Pq = f32(mm(Q, Wq.T, W.get("proj_q", W["proj"])) + bq)
        C = f32(torch.tanh(mm(Q, V_phys, W["affinity"])))
        Ct = C.transpose(1, 2)
        H_v = torch.tanh(Pv + mm(Ct, Pq, W["h"]))
        H_q = f32(torch.tanh(Pq + mm(C, Pv, W["h"])))
        a_v = f32(torch.softmax(H_v @ wv + cv, dim=1))
        a_q = f32(torch.softmax(H_q @ wq + cq, dim=1))
        res["v"].append((a_v.unsqueeze(2) * Vn).sum(1)); res["q"].append((a_q.unsqueeze(2) * Q).sum(1))
        res["C"].append(C); res["a_v"].append(a_v); res["a_q"].append(a_q); res["H_q"].append(H_q)
        # backward
        H_vb = torch.tanh(Pv + mm(Ct, Pq, W["bwd"]))                 # recomputed by both big kernels
        da_v = (Vn @ gv[l].double().unsqueeze(2)).squeeze(2)
        da_q = (Q @ gq[l].double().unsqueeze(2)).squeeze(2)
        dVn += a_v.unsqueeze(2) * gv[l].double().unsqueeze(1)
        ds_v = f32(a_v * (da_v - (a_v * da_v).sum(1, keepdim=True)))
        ds_q = a_q * (da_q - (a_q * da_q).sum(1, keepdim=True))
        g["w_v.weight"] += torch.einsum("bn,bnd->d", ds_v, H_vb).unsqueeze(0)
        g["w_q.weight"] += torch.einsum("bt,btd->d", ds_q, H_q).unsqueeze(0)
        g["w_v.bias"] += ds_v.sum().reshape(1)
        g["w_q.bias"] += ds_q.sum().reshape(1)
        dZ_v = f32(ds_v.unsqueeze(2) * wv * (1.0 - H_vb * H_vb))
        dZ_q = f32(ds_q.unsqueeze(2) * wq * (1.0 - H_q * H_q))
        dPv = f32(dZ_v + mm(Ct, dZ_q, W["bwd"]))
        dPq = f32(dZ_q + mm(C, dZ_v, W["bwd"]))
        dC = mm(Pq * wv, (dZ_v / wv).transpose(1, 2), W["bwd"]) + mm(dZ_q, Pv.transpose(1, 2), W["bwd"])
        dA = f32(dC * (1.0 - C * C))
        dQ = a_q.unsqueeze(2) * gq[l].double().unsqueeze(1) + mm(dA, Vn, W["dq"]) + mm(dPq, Wq, W["gemm_bwd"])
        dVn += mm(dA.transpose(1, 2), Q, W["dv"])
        dPv_tot += dPv
        g["W_q.weight"] += mm(dPq.reshape(-1, dPq.shape[-1]).T, Q.reshape(-1, Q.shape[-1]), W["gemm_bwd"])
        g["W_q.bias"] += dPq.sum((0, 1))
        dQs.append(dQ)
    dPv_tot = f32(dPv_tot)
    dVn += mm(dPv_tot, Wv, W["dv"])
    g["W_v.weight"] += mm(dPv_tot.reshape(-1, Pv.shape[-1]).T, Vn.reshape(-1, Vn.shape[-1]), W["gemm_bwd"])
    g["W_v.bias"] += dPv_tot.sum((0, 1))
    out = {k: torch.stack(v) for k, v in res.items()}
    out["dV_phys"] = dVn.permute(0, 2, 1).contiguous()
    out["dQ"] = torch.stack(dQs)
    out.update({"d" + k: v for k, v in g.items()})
    return out


def run_case(name, W):
    V, Qs, P, gv, gq = G.build_case(name, torch.float32)
    gold = G.load(name)
    r = emulate(V, Qs, P, gv, gq, W)
    return G.fwd_errors(r, gold), G.grad_errors(r, gold)


@pytest.mark.parametrize("name", sorted(G.CASES))
def test_mixed_widths_hold_the_contract(name):
    ef, eg = run_case(name, WIDTHS)
    worst_f, worst_g = max(ef.values()), max(eg.values())
    print("%s: forward %.2e  gradients %.2e  (%s)" % (name, worst_f, worst_g, max(eg, key=eg.get)))
    assert worst_f < 1e-4, ef
    assert worst_g < 1e-4, eg


def test_exact_split_is_at_fp32_level():
    ef, eg = run_case("g2_cfg2_natural", EXACT)
    assert max(ef.values()) < 5e-6 and max(eg.values()) < 5e-6, (ef, eg)


# the widths of rounds 3 / 4 before the FP16 pieces: the exact bf16 split where two bf16 pieces do not hold
BF16_WIDTHS = dict(WIDTHS, affinity=3, proj=3, proj_q=2, h=2)


def test_projections_do_not_hold_on_two_bf16_pieces():
    """Why the forward projections are not on two BF16 pieces: their error reaches H_q = tanh(P_q + C P_v) amplified by
    the sum over the N locations (2e-4 on the saturated case, past the contract) -- 16 significand bits are too few."""
    ef, _ = run_case("g2_cfg2_natural", dict(BF16_WIDTHS, proj=2))
    assert max(ef.values()) > 1e-4, ef


def test_affinity_does_not_hold_on_two_bf16_pieces():
    """The same for phase 1: two bf16 pieces in A = Q V^T break the gradient contract on the saturated case."""
    _, eg = run_case("g2_cfg2_natural", dict(BF16_WIDTHS, affinity=2))
    assert max(eg.values()) > 1e-4, eg


def test_fp16_pieces_beat_the_bf16_widths_they_replace():
    """Two FP16 pieces for the forward-side contractions (22 bits, half the MFMAs of the exact bf16 split): less error than
    exact affinity / P_v next to two-bf16-piece P_q / phase 2."""
    ef_h, eg_h = run_case("g2_cfg2_natural", WIDTHS)
    ef_b, eg_b = run_case("g2_cfg2_natural", BF16_WIDTHS)
    assert max(ef_h.values()) < 0.5 * max(ef_b.values()), (ef_h, ef_b)
    assert max(eg_h.values()) < max(eg_b.values()), (eg_h, eg_b)


def test_weight_scale_matters_for_fp16_pieces():
    """Without the factor 256 on the weight image the lo pieces of the weights are fp16 subnormals (2^-24 absolute): the
    projections then carry the largest error of the forward."""
    ef, _ = run_case("g2_cfg2_natural", dict(WIDTHS, proj="2h", proj_q="2h"))
    ef_s, _ = run_case("g2_cfg2_natural", WIDTHS)
    assert max(ef.values()) > 2 * max(ef_s.values()), (ef, ef_s)


if __name__ == "__main__":
    import sys
    tables = {"mixed": WIDTHS, "exact": EXACT}
    for k in ("proj", "h", "bwd", "dq", "gemm_bwd"):
        tables["only_" + k] = dict(EXACT, **{k: 2})
    tables["no_proj"] = dict(WIDTHS, proj=3)
    tables["no_proj_h"] = dict(WIDTHS, proj=3, h=3)
    tables["projq2"] = dict(WIDTHS, proj_q=2)
    tables["projv2x3"] = dict(WIDTHS, proj="2x3")
    tables["projv2x3_only"] = dict(EXACT, proj="2x3", proj_q=3)
    tables["fp16_fwd"] = dict(WIDTHS, proj="2h", h="2h")
    tables["fp16_fwd_only"] = dict(EXACT, proj="2h", h="2h")
    tables["fp16_all"] = {k: "2h" for k in WIDTHS}
    for tname, W in tables.items():
        if len(sys.argv) > 1 and tname not in sys.argv[1:]:
            continue
        for name in sorted(G.CASES):
            ef, eg = run_case(name, W)
            print("%-14s %-18s fwd %.2e (%s)  grad %.2e (%s)" % (tname, name, max(ef.values()), max(ef, key=ef.get),
                                                                 max(eg.values()), max(eg, key=eg.get)))
