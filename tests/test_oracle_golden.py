"""CPU: the oracle restatement (forward + hand-derived backward) against the committed golden
vectors, which were produced by the imported reference (oracle/make_golden.py)."""
import pytest
import torch

from oracle import coattn_oracle as O
from tests import _golden as G


@pytest.mark.parametrize("name", sorted(G.CASES))
def test_oracle_matches_reference_golden(name):
    gold = G.load(name)
    V, Qs, P, gv, gq = G.build_case(name, torch.float32)
    f = O.coattn_forward(V, Qs, P)
    g = O.coattn_backward(V, Qs, P, gv, gq)
    fe = G.fwd_errors(f, gold, "32")
    ge = G.grad_errors(g, gold, "32")
    assert max(fe.values()) < 2e-5, fe          # fp32 CPU vs reference fp32 CPU
    assert max(ge.values()) < 5e-5, ge
    fe64 = G.fwd_errors(f, gold, "64")
    assert max(fe64.values()) < 5e-5, fe64


def test_as_executed_equals_deduplicated():
    """De-duplicating the reference's 6x W_v(V) (model.py:380-384) is value preserving."""
    V, Qs, P, _, _ = G.build_case("g1_tiny_d64", torch.float64)
    a = O.coattn_forward(V, Qs, P, as_executed=True)
    b = O.coattn_forward(V, Qs, P, as_executed=False)
    for k in ("v", "q", "a_v", "a_q"):
        assert (a[k] - b[k]).abs().max().item() < 1e-13


def test_unmasked_softmax_quirk():
    """len=1 sample: most of a_q's mass lands on pad rows (model.py:388 has no mask)."""
    gold = G.load("g1_tiny_d64")
    aq = gold["a_q64"]                # [L,B,T]; sample 2 has len 1
    assert aq[:, 2, 1:].sum(-1).min() > 0.5
    assert abs(aq.sum(-1) - 1.0).max() < 1e-12


def test_manual_backward_matches_autograd_f64():
    V, Qs, P, gv, gq = G.build_case("g1_odd_d96", torch.float64)
    V = V.requires_grad_(True)
    Qs = [q.requires_grad_(True) for q in Qs]
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    f = O.coattn_forward(V, Qs, P)
    ((f["v"] * gv).sum() + (f["q"] * gq).sum()).backward()
    g = O.coattn_backward(V.detach(), [q.detach() for q in Qs], {k: v.detach() for k, v in P.items()}, gv, gq)
    assert (g["dV_phys"] - V.grad).abs().max() < 1e-12
    for l in range(3):
        assert (g["dQ"][l] - Qs[l].grad).abs().max() < 1e-12
    for k in O.PARAM_KEYS:
        assert (g["d" + k] - P[k].grad).abs().max() < 1e-11, k
    assert P["W_b.weight"].grad is None
