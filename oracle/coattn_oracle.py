"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's parallel co-attention.

Restates, in plain torch-on-CPU, what ``/root/reference/model.py`` computes on the hot
path (``ParallelCoAttention.forward``, model.py:356-397) plus a hand-derived backward
(the reference has no explicit backward: it is autograd of model.py:372-392, triggered at
main.py:219-220).  Parity status: PINNED by ``oracle/make_golden.py`` against the imported
reference classes (forward and every gradient), see ``tests/golden/MANIFEST.json``.

Layout conventions used everywhere in this repo (they are the reference's own):
  V_phys : [B, d, N]  physical image-feature buffer (NCHW flattened, model.py:215);
           the module-level input ``x_img`` is its permuted view [B, N, d] (model.py:217).
  Q_l    : [B, T, d]  contiguous, one per level (word / phrase / sentence), model.py:298.
  outputs: v_l, q_l : [B, d]  (model.py:391-392).

Reference behaviours that are restated on purpose:
  * ``W_b`` exists but is never used: C = tanh(Q V^T)            (model.py:347, :377)
  * Linear layers carry biases, row-vector form x W^T + b         (model.py:350-354)
  * one weight set shared by the three levels                     (model.py:167, :372)
  * softmax over all T positions, no padding mask                 (model.py:388)
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

PARAM_KEYS = ("W_v.weight", "W_v.bias", "W_q.weight", "W_q.bias",
              "w_v.weight", "w_v.bias", "w_q.weight", "w_q.bias")


# --------------------------------------------------------------------------------------
# deterministic closed-form tensors (so that fixtures store outputs only)
# --------------------------------------------------------------------------------------
def hash_uniform(n: int, seed: int) -> np.ndarray:
    """n uniforms in [0,1) from a 64-bit integer mix (splitmix64 finaliser); numpy only."""
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = z + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def hash_normal(shape: Sequence[int], seed: int, scale: float = 1.0) -> np.ndarray:
    """Approximately N(0, scale^2): sum of 4 hash uniforms, centred and rescaled (float64)."""
    n = int(np.prod(shape))
    u = hash_uniform(4 * n, seed).reshape(4, n).sum(0)
    return ((u - 2.0) * math.sqrt(3.0) * scale).reshape(shape)


def hash_unit(shape: Sequence[int], seed: int, bound: float) -> np.ndarray:
    """U(-bound, bound) (float64)."""
    n = int(np.prod(shape))
    return ((hash_uniform(n, seed) * 2.0 - 1.0) * bound).reshape(shape)


def make_params(d: int, seed: int, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Co-attention parameters with nn.Linear's default scale U(+-1/sqrt(d)) (model.py:347-354),
    generated in closed form.  Includes the dead ``W_b`` so state_dicts are complete."""
    bd = 1.0 / math.sqrt(d)
    p = {
        "W_b.weight": hash_unit((d, d), seed + 1, bd), "W_b.bias": hash_unit((d,), seed + 2, bd),
        "W_v.weight": hash_unit((d, d), seed + 3, bd), "W_v.bias": hash_unit((d,), seed + 4, bd),
        "W_q.weight": hash_unit((d, d), seed + 5, bd), "W_q.bias": hash_unit((d,), seed + 6, bd),
        "w_v.weight": hash_unit((1, d), seed + 7, bd), "w_v.bias": hash_unit((1,), seed + 8, bd),
        "w_q.weight": hash_unit((1, d), seed + 9, bd), "w_q.bias": hash_unit((1,), seed + 10, bd),
    }
    return {k: torch.from_numpy(v).to(dtype) for k, v in p.items()}


def make_inputs(B: int, N: int, T: int, d: int, seed: int, lens: Sequence[int] | None = None,
                scale_v: float = 1.0, scale_q: float = 1.0, relu_v: bool = True,
                dtype=torch.float32, L: int = 3):
    """Synthetic features in the layouts of the path: V_phys [B,d,N] (ReLU-like, >= 0, as a
    VGG feature map is, model.py:212) and L question tensors [B,T,d] whose rows t >= len are
    exact zeros (padding_idx=0 model.py:263; zero-fill of pad_packed_sequence model.py:292-296)."""
    V = hash_normal((B, d, N), seed + 100, scale_v)
    if relu_v:
        V = np.maximum(V, 0.0)
    Qs = [hash_normal((B, T, d), seed + 200 + l, scale_q) for l in range(L)]
    if lens is not None:
        for b, ln in enumerate(lens):
            for q in Qs:
                q[b, int(ln):, :] = 0.0
    return (torch.from_numpy(V).to(dtype),
            [torch.from_numpy(q).to(dtype) for q in Qs])


# --------------------------------------------------------------------------------------
# forward (functional), as the reference computes it -- model.py:372-392
# --------------------------------------------------------------------------------------
def coattn_forward(V_phys: torch.Tensor, Qs: List[torch.Tensor], P: Dict[str, torch.Tensor],
                   as_executed: bool = False) -> Dict[str, torch.Tensor]:
    """Returns v,q [L,B,d] and the intermediates C [L,B,T,N], a_v [L,B,N], a_q [L,B,T],
    H_q [L,B,T,d].

    ``as_executed=True`` re-evaluates W_v(V) and W_q(Q) twice per level exactly as
    model.py:380-384 does (6x W_v(V) per step); False evaluates W_v(V) once per sample.  The
    two are value-identical up to rounding (SURVEY.md section 8 probe)."""
    Wv, bv, Wq, bq = P["W_v.weight"], P["W_v.bias"], P["W_q.weight"], P["W_q.bias"]
    wv, cv, wq, cq = P["w_v.weight"], P["w_v.bias"], P["w_q.weight"], P["w_q.bias"]
    Vn = V_phys.permute(0, 2, 1)                       # [B,N,d] view  (model.py:217 / :378)
    out = {k: [] for k in ("v", "q", "C", "a_v", "a_q", "H_q")}
    Pv_once = None if as_executed else F.linear(Vn, Wv, bv)
    for Q in Qs:                                       # model.py:372 (shared weights)
        C = torch.tanh(torch.bmm(Q, V_phys))           # model.py:377   [B,T,N]
        if as_executed:
            H_v = torch.tanh(F.linear(Vn, Wv, bv) + torch.bmm(C.transpose(2, 1), F.linear(Q, Wq, bq)))
            H_q = torch.tanh(F.linear(Q, Wq, bq) + torch.bmm(C, F.linear(Vn, Wv, bv)))
        else:
            Pq = F.linear(Q, Wq, bq)
            H_v = torch.tanh(Pv_once + torch.bmm(C.transpose(2, 1), Pq))    # model.py:380-381
            H_q = torch.tanh(Pq + torch.bmm(C, Pv_once))                    # model.py:383-384
        a_v = F.softmax(F.linear(H_v, wv, cv), dim=1)   # model.py:387   [B,N,1]
        a_q = F.softmax(F.linear(H_q, wq, cq), dim=1)   # model.py:388   [B,T,1]  (no mask)
        out["v"].append(torch.sum(a_v * Vn, dim=1))     # model.py:391
        out["q"].append(torch.sum(a_q * Q, dim=1))      # model.py:392
        out["C"].append(C); out["a_v"].append(a_v.squeeze(2)); out["a_q"].append(a_q.squeeze(2))
        out["H_q"].append(H_q)
    return {k: torch.stack(v) for k, v in out.items()}


# --------------------------------------------------------------------------------------
# hand-derived backward ("Backward"); checked against autograd of the
# imported reference in make_golden.py and tests/test_oracle.py
# --------------------------------------------------------------------------------------
def coattn_backward(V_phys: torch.Tensor, Qs: List[torch.Tensor], P: Dict[str, torch.Tensor],
                    gv: torch.Tensor, gq: torch.Tensor) -> Dict[str, torch.Tensor]:
    """gv,gq: [L,B,d] upstream gradients of v_l,q_l.  Returns dV_phys [B,d,N], dQ [L,B,T,d] and
    the parameter gradients under the reference's state_dict names (W_b has none)."""
    Wv, bv, Wq, bq = P["W_v.weight"], P["W_v.bias"], P["W_q.weight"], P["W_q.bias"]
    wv, cv, wq, cq = P["w_v.weight"][0], P["w_v.bias"], P["w_q.weight"][0], P["w_q.bias"]
    Vn = V_phys.permute(0, 2, 1)                                    # [B,N,d]
    Pv = Vn @ Wv.T + bv
    dVn = torch.zeros_like(Vn)
    g = {k: torch.zeros_like(P[k]) for k in PARAM_KEYS}
    dQs = []
    dPv_tot = torch.zeros_like(Pv)
    for l, Q in enumerate(Qs):
        Pq = Q @ Wq.T + bq
        C = torch.tanh(Q @ V_phys)                                  # [B,T,N]
        H_v = torch.tanh(Pv + C.transpose(1, 2) @ Pq)               # [B,N,d]
        H_q = torch.tanh(Pq + C @ Pv)                               # [B,T,d]
        a_v = torch.softmax(H_v @ wv + cv, dim=1)                   # [B,N]
        a_q = torch.softmax(H_q @ wq + cq, dim=1)                   # [B,T]
        # v = a_v^T V ; q = a_q^T Q
        da_v = Vn @ gv[l].unsqueeze(2)                              # [B,N,1]
        da_q = Q @ gq[l].unsqueeze(2)                               # [B,T,1]
        dVn += a_v.unsqueeze(2) * gv[l].unsqueeze(1)
        dQ = a_q.unsqueeze(2) * gq[l].unsqueeze(1)
        ds_v = a_v * (da_v.squeeze(2) - (a_v * da_v.squeeze(2)).sum(1, keepdim=True))
        ds_q = a_q * (da_q.squeeze(2) - (a_q * da_q.squeeze(2)).sum(1, keepdim=True))
        g["w_v.weight"] += torch.einsum("bn,bnd->d", ds_v, H_v).unsqueeze(0)
        g["w_q.weight"] += torch.einsum("bt,btd->d", ds_q, H_q).unsqueeze(0)
        g["w_v.bias"] += ds_v.sum().reshape(1)
        g["w_q.bias"] += ds_q.sum().reshape(1)
        dZ_v = ds_v.unsqueeze(2) * wv * (1.0 - H_v * H_v)           # [B,N,d]
        dZ_q = ds_q.unsqueeze(2) * wq * (1.0 - H_q * H_q)           # [B,T,d]
        dPv = dZ_v + C.transpose(1, 2) @ dZ_q
        dPq = dZ_q + C @ dZ_v
        dC = Pq @ dZ_v.transpose(1, 2) + dZ_q @ Pv.transpose(1, 2)  # [B,T,N]
        dA = dC * (1.0 - C * C)
        dQ = dQ + dA @ Vn + dPq @ Wq
        dVn += dA.transpose(1, 2) @ Q
        dPv_tot += dPv
        g["W_q.weight"] += torch.einsum("btj,btk->jk", dPq, Q)
        g["W_q.bias"] += dPq.sum((0, 1))
        dQs.append(dQ)
    dVn += dPv_tot @ Wv
    g["W_v.weight"] += torch.einsum("bnj,bnk->jk", dPv_tot, Vn)
    g["W_v.bias"] += dPv_tot.sum((0, 1))
    out = {"dV_phys": dVn.permute(0, 2, 1).contiguous(), "dQ": torch.stack(dQs)}
    out.update({"d" + k: v for k, v in g.items()})
    return out


# --------------------------------------------------------------------------------------
# nn.Module restatements (state_dict keys = the reference's; used for the CPU baseline and
# for autograd cross-checks)
# --------------------------------------------------------------------------------------
class OracleParallelCoAttention(nn.Module):
    """Restates model.py:337-397 with the same attribute names and call signature."""

    def __init__(self, hidden_dim: int, as_executed: bool = True):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.as_executed = as_executed
        self.W_b = nn.Linear(hidden_dim, hidden_dim)     # dead, model.py:347
        self.W_v = nn.Linear(hidden_dim, hidden_dim)
        self.W_q = nn.Linear(hidden_dim, hidden_dim)
        self.w_v = nn.Linear(hidden_dim, 1)
        self.w_q = nn.Linear(hidden_dim, 1)

    def forward(self, x_img, x_ques_hierarchy):
        P = {k: v for k, v in self.named_parameters()}
        r = coattn_forward(x_img.permute(0, 2, 1), list(x_ques_hierarchy), P, self.as_executed)
        L = r["v"].shape[0]
        return [r["v"][l] for l in range(L)], [r["q"][l] for l in range(L)]


class OracleMLPClassifier(nn.Module):
    """Restates model.py:400-434 (recursive word -> phrase -> sentence encoding)."""

    def __init__(self, hidden_dim: int, mlp_dim: int, K: int):
        super().__init__()
        self.W_w = nn.Linear(hidden_dim, hidden_dim)
        self.W_p = nn.Linear(2 * hidden_dim, hidden_dim)
        self.W_s = nn.Linear(2 * hidden_dim, mlp_dim)
        self.W_h = nn.Linear(mlp_dim, K)

    def forward(self, x_img_feats, x_ques_feats):
        q_w, q_p, q_s = x_ques_feats
        v_w, v_p, v_s = x_img_feats
        h_w = torch.tanh(self.W_w(q_w + v_w))
        h_p = torch.tanh(self.W_p(torch.cat([q_p + v_p, h_w], dim=1)))
        h_s = torch.tanh(self.W_s(torch.cat([q_s + v_s, h_p], dim=1)))
        return self.W_h(h_s)


def checksum(t: torch.Tensor, seed: int = 7, nsamp: int = 64) -> Dict[str, np.ndarray]:
    """Size-independent summary of a tensor for compact golden files: sum, sum|x|, three fixed
    pseudo-random linear functionals and ``nsamp`` sampled entries (all float64)."""
    x = t.detach().to(torch.float64).reshape(-1).numpy()
    n = x.size
    idx = (hash_uniform(nsamp, seed + 17) * n).astype(np.int64)
    proj = [float(x @ hash_unit((n,), seed + 31 + j, 1.0)) for j in range(3)]
    return {"n": np.int64(n), "sum": np.float64(x.sum()), "abs": np.float64(np.abs(x).sum()),
            "proj": np.array(proj), "idx": idx, "samp": x[idx]}
