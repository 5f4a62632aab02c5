"""TEST INFRASTRUCTURE ONLY -- the table of golden cases for the co-attention path and the
closed-form regeneration of their inputs (so that ``tests/golden/*.npz`` hold outputs only).

Case ids follow SURVEY.md section 8c (G1..G6); every case pins forward outputs, level-0
intermediates and every gradient of ``ParallelCoAttention`` (model.py:337-397) produced by
the imported reference in float32 (the reference's CPU path) and in float64.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from . import coattn_oracle as O

# name -> dict(B,N,T,d, lens, scale_v, scale_q, seed, full)   (full: store whole tensors)
CASES: Dict[str, dict] = {
    # G1 tiny, ragged lengths incl. len=1 (96.5 % of a_q's mass lands on pad rows)
    "g1_tiny_d64": dict(B=3, N=49, T=26, d=64, lens=[26, 9, 1], scale_v=0.5, scale_q=0.5, seed=101, full=True),
    # odd shapes for the general-shape kernels (N,T,d not multiples of the tile sizes)
    "g1_odd_d96": dict(B=2, N=37, T=5, d=96, lens=[5, 2], scale_v=0.4, scale_q=0.4, seed=102, full=True),
    # G2 headline kernel shape (448x448 -> 14x14 grid), natural scale: tanh saturates
    "g2_cfg2_natural": dict(B=2, N=196, T=26, d=512, lens=[26, 7], scale_v=1.0, scale_q=1.0, seed=103, full=False),
    # G5 same shape, 1/sqrt(d)-scaled: unsaturated tanh, exercises the dA path
    "g5_cfg2_scaled": dict(B=2, N=196, T=26, d=512, lens=[26, 7], scale_v=1.0,
                           scale_q=math.sqrt(2.0 / 512), seed=104, full=False),
    # G3/G4 224x224 -> 7x7 grid, ragged incl. all-26 and len=1
    "g3_n49_ragged": dict(B=3, N=49, T=26, d=512, lens=[26, 26, 1], scale_v=1.0,
                          scale_q=math.sqrt(2.0 / 512), seed=105, full=False),
    # d=256 (two-wave configuration of the fused kernels)
    "g5_d256": dict(B=2, N=196, T=26, d=256, lens=[20, 3], scale_v=1.0,
                    scale_q=math.sqrt(2.0 / 256), seed=106, full=False),
    # G6 ResNet-like 7x7x2048 features (fp32 oracle for BASELINE config 4)
    "g6_d2048_n49": dict(B=2, N=49, T=26, d=2048, lens=[26, 11], scale_v=1.0,
                         scale_q=math.sqrt(2.0 / 2048), seed=107, full=False),
}


def build_case(name: str, dtype=torch.float32):
    """Regenerates (V_phys, Qs, params, gv, gq) of a case in closed form."""
    c = CASES[name]
    P = O.make_params(c["d"], c["seed"], dtype)
    V, Qs = O.make_inputs(c["B"], c["N"], c["T"], c["d"], c["seed"], lens=c["lens"],
                          scale_v=c["scale_v"], scale_q=c["scale_q"], dtype=dtype)
    gv = torch.from_numpy(O.hash_normal((3, c["B"], c["d"]), c["seed"] + 900)).to(dtype)
    gq = torch.from_numpy(O.hash_normal((3, c["B"], c["d"]), c["seed"] + 901)).to(dtype)
    return V, Qs, P, gv, gq


GRAD_KEYS = ("dV_phys", "dQ") + tuple("d" + k for k in O.PARAM_KEYS)


# ---- network-level cases (G7 logits, G8 train-step trajectory, G9 baseline net) -----------
NET_CASE = dict(vocab=40, hidden=512, K=10, B=3, T=8, image=64, lens=[8, 5, 2], seed=301, lr=1e-4, steps=3)
BASE_CASE = dict(vocab=30, emb=300, hidden=1024, K=2, B=2, T=6, lens=[6, 3], seed=401)


def closed_form_state(module, seed: int):
    """Closed-form values for every entry of ``module.state_dict()`` (in its own order):
    weights U(+-sqrt(3/fan_in)), biases U(+-0.05), BatchNorm affine ~1 / running stats at their
    defaults, embeddings ~N(0,1) with row 0 = 0 (padding_idx).  Returns a dict of tensors."""
    import numpy as np

    out = {}
    for i, (name, t) in enumerate(module.state_dict().items()):
        s = seed * 1000 + i
        if name.endswith("num_batches_tracked"):
            v = torch.zeros_like(t)
        elif name.endswith("running_mean"):
            v = torch.zeros_like(t)
        elif name.endswith("running_var"):
            v = torch.ones_like(t)
        elif "word_embedding" in name and t.dim() == 2:
            v = torch.from_numpy(O.hash_normal(tuple(t.shape), s)).to(t.dtype)
            v[0] = 0
        elif t.dim() == 1:
            base = 1.0 if (name.endswith("weight")) else 0.0          # BatchNorm gamma ~ 1
            v = torch.from_numpy(base + O.hash_unit(tuple(t.shape), s, 0.05)).to(t.dtype)
        else:
            fan_in = int(np.prod(t.shape[1:]))
            v = torch.from_numpy(O.hash_unit(tuple(t.shape), s, math.sqrt(3.0 / fan_in))).to(t.dtype)
        out[name] = v
    return out


def net_case_batch(c=None):
    """(image, question, ques_len, label) of the network-level case, sorted by length (desc)."""
    c = c or NET_CASE
    B, T = c["B"], c["T"]
    image = torch.from_numpy(O.hash_normal((B, 3, c["image"], c["image"]), c["seed"] + 1)).float()
    tok = (O.hash_uniform(B * T, c["seed"] + 2) * (c["vocab"] - 2)).astype("int64").reshape(B, T) + 2
    question = torch.from_numpy(tok)
    lens = torch.tensor(c["lens"], dtype=torch.int64)
    question = question * (torch.arange(T)[None, :] < lens[:, None])
    label = torch.from_numpy((O.hash_uniform(B, c["seed"] + 3) * (c["K"] + 1)).astype("int64"))
    return image, question, lens, label


# ---- network-level cases with INJECTED image features (SURVEY 8c G7 "stub VGG features injected") ---------
# The HIP path is then compared inside the network at the op-level tolerance (1e-4): nothing stock and
# batch-statistics dependent (MIOpen convolutions, 3-sample BatchNorm) sits in front of it.
#   NETF_CASE: cfg-2 like (hidden 512, 7x7 grid).   NETF4_CASE: BASELINE config 4 (ResNet-152-like 7x7x2048
#   features, hidden 2048, K=3000 -> 3001 logits), fp32 reference values; the bf16 mode is checked against
#   them at a stated bf16 tolerance.
NETF_CASE = dict(vocab=40, hidden=512, K=10, B=3, T=26, N=49, lens=[26, 11, 3], seed=501, lr=1e-4, steps=3)
NETF4_CASE = dict(vocab=30, hidden=2048, K=3000, B=2, T=26, N=49, lens=[26, 9], seed=601, lr=1e-4, steps=2)


def netf_inputs(c):
    """(features [B,N,d] as the permuted view of a channel-major [B,d,N] buffer -- the reference's layout,
    model.py:215-217 --, question, ques_len, label), closed form."""
    B, T, N, d = c["B"], c["T"], c["N"], c["hidden"]
    feats = torch.from_numpy(O.hash_normal((B, d, N), c["seed"] + 1)).float().clamp_min(0).permute(0, 2, 1)
    tok = (O.hash_uniform(B * T, c["seed"] + 2) * (c["vocab"] - 2)).astype("int64").reshape(B, T) + 2
    lens = torch.tensor(c["lens"], dtype=torch.int64)
    question = torch.from_numpy(tok) * (torch.arange(T)[None, :] < lens[:, None])
    label = torch.from_numpy((O.hash_uniform(B, c["seed"] + 3) * (c["K"] + 1)).astype("int64"))
    return feats, question, lens, label
